#!/usr/bin/env python3
"""CPU-side timing of a Bayes-tree pass (TEST INFRASTRUCTURE; bench.py's cpu_baseline leg runs this in a fresh process).

    python oracle/cpu_tree_bench.py manhattan.g2o --poses 200 [--messages relative] [--particles 100]

The oracle's restatement of the tree solve -- the schedule of rome_jl_amd/tree.py with every convolution, `manikde!` bandwidth, multiscale
Gibbs product and block operation computed by oracle/ (C) through the Python stand-in backend of tests/dist_standin.py -- on the
sub-graph of the first `poses` poses of the g2o file (odometry and the loop closures among them).  The stand-in walks the rows of a level
one by one from Python; the threads the oracle's OpenMP loops use inside a call are reported.  Prints one JSON object: seconds of one up + down pass, tree statistics.  It is a bounded SAMPLE
of the workload, not the 3500-pose graph (that pass takes minutes here)."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("g2o")
    ap.add_argument("--poses", type=int, default=200)
    ap.add_argument("--messages", default="relative")
    ap.add_argument("--particles", type=int, default=100)
    a = ap.parse_args()
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import rome_jl_amd as R
    from rome_jl_amd.tree import TreeSolver
    from dist_standin import OracleTreeBackend
    import oracle as ro
    N = a.particles
    fg = R.initfg(N)
    rows, ids = [], set()
    for ln in open(a.g2o):
        t = ln.split()
        if t and t[0] == "EDGE_SE2" and int(t[1]) < a.poses and int(t[2]) < a.poses:
            rows.append(t); ids.update((int(t[1]), int(t[2])))
    for k in sorted(ids):
        fg.addVariable("x%d" % k, R.Pose2)
    fg.addFactor(["x0"], R.PriorPose2(R.MvNormal(np.zeros(3), np.diag([0.01, 0.01, 0.0025]))))
    for t in rows:
        u = [float(x) for x in t[6:12]]
        C = np.linalg.inv(np.array([[u[0], u[1], u[2]], [u[1], u[3], u[4]], [u[2], u[4], u[5]]]))
        fg.addFactor(["x%s" % t[1], "x%s" % t[2]], R.Pose2Pose2(R.MvNormal(np.array([float(x) for x in t[3:6]]), 0.5 * (C + C.T))))
    R.dead_reckon_init(fg, seed=1)
    ts = TreeSolver(fg, messages=a.messages, backend=OracleTreeBackend(R))
    ts.upload()
    t0 = time.perf_counter()
    ts.solve(R.make_opts(N=N, seed=3))
    dt = time.perf_counter() - t0
    st = ts.stats()
    print(json.dumps({"poses": len(ids), "factors": len(fg.factors), "particles": N, "messages": a.messages, "seconds_per_pass": dt, "threads": ro.num_threads(), "rows_per_pass": st["up_rows"] + st["down_rows"], "levels": st["levels"], "cliques": st["cliques"]}))


if __name__ == "__main__":
    main()
