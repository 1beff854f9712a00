#!/bin/bash
# scripts/profile_round.sh <tag>   (on the GPU box, through gpurun)
# rocprofv3 evidence for profiles/: kernel traces of bench.py for the three solvers, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes + the copy8 calibration), SQ counters.  Raw rocprof output stays in /tmp; only summaries go to
# gpurun_out/<tag>/ (copy what is to be judged into profiles/).
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; T=/tmp/prof_$tag; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_TRACES" ]; then
# 1. kernel traces (same command as the bench line: defaults = 4000 settle+warm-up launches, 2000 timed)
for s in newton closed_form gauss_newton nelder_mead; do
  st=2000; wu=2000; [ $s = nelder_mead ] && st=20 && wu=10
  timeout 400 rocprofv3 --kernel-trace --stats -d $T/trace_$s -o $s -- python $R/bench.py --solver $s --steps $st --warmup $wu --no-cpu-baseline --no-modes > $T/trace_$s.log 2>&1
  db=$(find $T/trace_$s -name "*_results.db" | head -1)
  python3 $R/scripts/rocpd_summary.py $db $O/${tag}_kernel_trace_$s.md > /dev/null
  grep '^{' $T/trace_$s.log | tail -1 > $O/${tag}_bench_under_trace_$s.json
done
fi
# 2. HBM traffic
[ -x $R/scripts/ubench/copy8 ] || hipcc --offload-arch=gfx950 -O3 -o $R/scripts/ubench/copy8 $R/scripts/ubench/copy8.hip
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $T/cal_$c -o cal -- $R/scripts/ubench/copy8 > $T/cal_$c.log 2>&1
  for s in newton closed_form gauss_newton; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $T/${s}_$c -o $s -- python $R/bench.py --solver $s --steps 20 --warmup 2 --no-cpu-baseline --no-modes > $T/${s}_$c.log 2>&1
  done
done
# 3. SQ counters (Nelder-Mead too: its bound is FP64-VALU issue, the busy fraction is the roofline that applies)
for s in newton closed_form gauss_newton nelder_mead; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace -d $T/sqa_$s -o a -- python $R/bench.py --solver $s --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $T/sqa_$s.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace -d $T/sqb_$s -o b -- python $R/bench.py --solver $s --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $T/sqb_$s.log 2>&1
done
python3 - <<PY
import sqlite3, glob, json
T, O, tag = "$T", "$O", "$tag"
def counters(pattern, like):
    out = {}
    for d in sorted(glob.glob(pattern, recursive=True)):
        db = sqlite3.connect(d)
        for name, cn, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            if like in name: out[cn] = (avg, n, name)
        for name, avg in db.execute("select name, avg(duration) from kernels group by name"):
            if like in name: out.setdefault("_dur", []).append(avg)
    return out
cal = {c: counters("%s/cal_%s/**/*_results.db" % (T, c), "copy")[c][0] for c in ("FETCH_SIZE", "WRITE_SIZE")}
fetch_scale = 2.0 if cal["FETCH_SIZE"] < 0.75 * 2097152 else 1.0   # copy8 reads 2 GiB: gfx950 FETCH_SIZE reports half of it
ALG = {"newton": 10906 * 100 * 48 + 2400, "closed_form": 10906 * 100 * 48 + 2400, "gauss_newton": 10906 * 100 * 72 + 2400,
       "nelder_mead": 10906 * 100 * 72 + 2400}   # bench.py BYTES_PER_PARTICLE_P2P2
for s in ("newton", "closed_form", "gauss_newton", "nelder_mead"):
    if s != "nelder_mead":   # (Nelder-Mead: counters only -- 2.3 ms of arithmetic per 78 MB, no FETCH/WRITE pass)
        f = counters("%s/%s_FETCH_SIZE/**/*_results.db" % (T, s), "k_conv"); w = counters("%s/%s_WRITE_SIZE/**/*_results.db" % (T, s), "k_conv")
        rd = f["FETCH_SIZE"][0] * 1024 * fetch_scale; wr = w["WRITE_SIZE"][0] * 1024
        json.dump({"source": "scripts/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --solver %s --steps 20; "
                             "FETCH_SIZE x%.0f per the gfx950 correction, calibrated in the same run on scripts/ubench/copy8 (2 GiB read reported %.4g KB, 2 GiB written reported %.4g KB)"
                             % (s, fetch_scale, cal["FETCH_SIZE"], cal["WRITE_SIZE"]),
                   "solver": s, "n_conv": 10907, "kernel": f["FETCH_SIZE"][2], "fetch_size_kb_raw": f["FETCH_SIZE"][0], "write_size_kb_raw": w["WRITE_SIZE"][0],
                   "launches_averaged": f["FETCH_SIZE"][1], "bytes_per_launch": int(rd + wr), "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr),
                   "algorithmic_bytes_per_launch": ALG[s]}, open("%s/hbm_traffic_%s.json" % (O, s), "w"), indent=1)
    a = counters("%s/sqa_%s/**/*_results.db" % (T, s), "k_conv"); b = counters("%s/sqb_%s/**/*_results.db" % (T, s), "k_conv")
    dur = sum(a["_dur"]) / len(a["_dur"])
    waves = a["SQ_WAVES"][0]
    json.dump({"source": "scripts/profile_round.sh: rocprofv3 --pmc SQ_* --kernel-trace, bench.py --solver %s --steps 10 after the settle launches (averages over %d steady-state launches, Manhattan M3500), per launch of %s"
                         % (s, a["SQ_WAVES"][1], a["SQ_WAVES"][2]),
               "SQ_WAVES": waves, "SQ_INSTS_VALU": a["SQ_INSTS_VALU"][0], "SQ_ACTIVE_INST_VALU_quadcycles": a["SQ_ACTIVE_INST_VALU"][0],
               "SQ_WAVE_CYCLES_quadcycles": a["SQ_WAVE_CYCLES"][0], "SQ_WAIT_INST_ANY_quadcycles": a["SQ_WAIT_INST_ANY"][0], "SQ_WAIT_ANY_quadcycles": a["SQ_WAIT_ANY"][0],
               "SQ_INSTS_SALU": a["SQ_INSTS_SALU"][0], "SQ_INSTS_VMEM_RD": b["SQ_INSTS_VMEM_RD"][0], "SQ_INSTS_VMEM_WR": b["SQ_INSTS_VMEM_WR"][0],
               "SQ_INSTS_SMEM": b["SQ_INSTS_SMEM"][0], "GRBM_GUI_ACTIVE": b["GRBM_GUI_ACTIVE"][0], "kernel_ns": dur,
               "derived": {"clock_GHz": 2.4, "valu_instructions_per_wave": a["SQ_INSTS_VALU"][0] / waves,
                           "valu_busy_fraction": 4.0 * a["SQ_ACTIVE_INST_VALU"][0] / (256 * 4 * 2.4 * dur)}},
              open("%s/sq_counters_%s.json" % (O, s), "w"), indent=1)
PY
rm -rf $T
ls -la $O
