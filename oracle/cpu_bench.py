#!/usr/bin/env python3
"""CPU-reference timing of the factor-convolution path (TEST INFRASTRUCTURE; bench.py's cpu_baseline leg runs this in a
fresh process so that the OpenMP runtime is configured before it starts and no other thread pool competes).

    python oracle/cpu_bench.py table.npz [--budget S] [--threads T,...]

table.npz: mu (F,3), L (F,6), bel (V,3,N), factor/dir/fixed/target (C,) -- the Pose2Pose2 convolution table bench.py runs on
the GPU.  For each solver (Nelder-Mead = the reference's Optim algorithm, Newton, closed form) and each thread count the
oracle convolves a bounded prefix of the table >= 5 times and reports the MEDIAN rate (SURVEY.md 8(d) "CPU reference timing").
The oracle is built here with -O3 -march=native (oracle/Makefile `native`).  Prints one JSON object.
"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def host_info():
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q = f.read().split()
        info["cgroup_cpu_max"] = " ".join(q)
        if q[0] != "max":
            info["cgroup_cpus"] = float(q[0]) / float(q[1])
    except Exception:
        pass
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            k = line.split(":")[0].strip()
            if k in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "NUMA node(s)"):
                info[k] = line.split(":", 1)[1].strip()
    except Exception:
        pass
    return info


def worker(args):
    """one (thread count) measurement in THIS process: OMP_* already in the environment"""
    import numpy as np
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle as ro
    d = np.load(args.table)
    N = d["bel"].shape[2]
    T = ro.num_threads()
    C = len(d["factor"])
    res = {"threads": T}
    for name, solver, per_thread_guess in (("nelder_mead", ro.SOLVER_NELDER_MEAD, 200.0), ("newton", ro.SOLVER_NEWTON, 8000.0),
                                           ("closed_form", ro.SOLVER_CLOSED_FORM, 40000.0)):
        o = ro.make_opts(N=N, solver=solver, seed=0x524F4D45)

        def run(n):
            t = time.perf_counter()
            ro.conv_pose2pose2(o, d["mu"], d["L"], d["bel"], d["fixed"][:n], d["target"][:n], d["dir"][:n], factor=d["factor"][:n])
            return time.perf_counter() - t

        n = int(min(C, max(8 * T, per_thread_guess * T * 0.05)))
        t0 = run(n)                                         # calibrate (also warms the thread pool)
        n = int(min(C, max(8 * T, n * args.per_rep / max(t0, 1e-4))))
        times = [run(n) for _ in range(args.reps)]
        times.sort()
        med = times[len(times) // 2]
        res[name] = {"conv_per_s": n / med, "sample_convolutions": n, "reps": args.reps, "median_s": med, "min_s": times[0], "max_s": times[-1]}
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("table")
    ap.add_argument("--budget", type=float, default=24.0, help="seconds of CPU timing in total (approximately)")
    ap.add_argument("--threads", default=None, help="comma list of thread counts (default: 1 and all usable cores)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--per-rep", type=float, default=None)
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    info = host_info()
    usable = info["affinity"]
    if "cgroup_cpus" in info:
        usable = max(1, min(usable, int(info["cgroup_cpus"] + 0.5)))
    # physical cores only when SMT is on (two FP64-bound threads on one core share its FMA pipes)
    try:
        tpc = int(info.get("Thread(s) per core", "1"))
    except ValueError:
        tpc = 1
    threads = [int(x) for x in args.threads.split(",")] if args.threads else sorted({1, max(1, usable // tpc), usable})
    flags = "-O3 -march=native -fopenmp -ffp-contract=off"
    so = os.path.join(HERE, "librome_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", HERE, "-B", "librome_oracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        so = os.path.join(HERE, "librome_oracle.so"); flags = "-O2 -fopenmp -ffp-contract=off (native build failed)"
    per_rep = args.per_rep or args.budget / (len(threads) * 3 * (args.reps + 1))
    runs = []
    for T in threads:
        env = dict(os.environ, ROME_ORACLE_SO=so, OMP_NUM_THREADS=str(T), OMP_PROC_BIND="spread", OMP_PLACES="cores" if T <= usable // tpc else "threads",
                   OMP_DYNAMIC="false", OMP_WAIT_POLICY="active")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), args.table, "--worker", "--reps", str(args.reps), "--per-rep", str(per_rep)],
                           env=env, capture_output=True, text=True)
        if p.returncode != 0:
            raise SystemExit("cpu_bench worker failed: " + p.stderr[-2000:])
        runs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    one = next(r for r in runs if r["threads"] == 1) if any(r["threads"] == 1 for r in runs) else None
    best = {}
    for name in ("nelder_mead", "newton", "closed_form"):
        top = max(runs, key=lambda r: r[name]["conv_per_s"])
        best[name] = {"threads": top["threads"], "conv_per_s": top[name]["conv_per_s"],
                      "one_thread_conv_per_s": one[name]["conv_per_s"] if one else None,
                      "parallel_efficiency": (top[name]["conv_per_s"] / (top["threads"] * one[name]["conv_per_s"])) if one else None,
                      "sample_convolutions": top[name]["sample_convolutions"], "reps": top[name]["reps"], "median_s": top[name]["median_s"]}
    print(json.dumps({"host": info, "flags": flags, "omp": "OMP_PROC_BIND=spread OMP_PLACES=cores|threads OMP_WAIT_POLICY=active, fresh process per thread count",
                      "runs": runs, "best": best}))


if __name__ == "__main__":
    main()
