#!/bin/bash
# kernel trace + SQ counters of the elimination solve of Manhattan-3500 + per-step wall-clock
#   scripts/elimination_trace.sh [tag] -> gpurun_out/<tag>_elimination_kernel_trace.md, <tag>_elimination_steps.txt, <tag>_elimination_sq_counters.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=/tmp/elim_trace; TAG=${1:-r06}; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/scripts/elimination_passes.py --passes 20 --steps --out $O/${TAG}_elimination_steps.txt > $T/steps.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $T/out -o g -- python $R/scripts/elimination_passes.py --passes 20 > $T/log.txt 2>&1
python3 $R/scripts/rocpd_summary.py $(find $T/out -name "*_results.db" | head -1) $O/${TAG}_elimination_kernel_trace.md > /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $T/a -o a -- python $R/scripts/elimination_passes.py --passes 4 > $T/a.log 2>&1
python3 - <<PY
import sqlite3, glob, json
T, O, tag = "$T", "$O", "$TAG"
res = {}
KS = ("k_kde_bandwidth_fast", "k_product_gibbs", "k_gibbs_trees", "k_conv<", "k_conv_flat", "k_block_ops", "k_scatter_blocks")
for d in sorted(glob.glob("%s/a/**/*_results.db" % T, recursive=True)):
    db = sqlite3.connect(d)
    try:
        rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        rows = []; print("no counters:", e)
    for name, cn, n, avg, tot in rows:
        for k in KS:
            if k in name:
                r = res.setdefault(k, {"kernel": name[:90]}); r[cn + "_sum"] = r.get(cn + "_sum", 0) + tot; r["launches_" + cn] = r.get("launches_" + cn, 0) + n
    for name, n, tot in db.execute("select name, count(*), sum(duration) from kernels group by name"):
        for k in KS:
            if k in name:
                r = res.setdefault(k, {"kernel": name[:90]}); r["total_ns"] = r.get("total_ns", 0) + tot; r["launches"] = r.get("launches", 0) + n
for k, r in res.items():
    if "SQ_WAVES_sum" in r and r.get("total_ns"):
        r["derived"] = {"waves_per_launch": r["SQ_WAVES_sum"] / r["launches"], "valu_instructions_per_wave": r["SQ_INSTS_VALU_sum"] / max(r["SQ_WAVES_sum"], 1),
                        "avg_us_per_launch": r["total_ns"] / r["launches"] / 1e3,
                        "chip_valu_busy_fraction": 4.0 * r["SQ_ACTIVE_INST_VALU_sum"] / (256 * 4 * 2.4 * r["total_ns"])}
json.dump({"source": "scripts/elimination_trace.sh: rocprofv3 --pmc SQ_* --kernel-trace over 5 elimination passes of Manhattan-3500, N = 100; sums over all launches of a "
                     "kernel; chip_valu_busy_fraction = VALU-busy SIMD cycles / (1024 SIMDs x kernel time)", "kernels": res},
          open("%s/%s_elimination_sq_counters.json" % (O, tag), "w"), indent=1)
for k, r in res.items():
    print(k, r.get("launches"), r.get("derived"))
PY
cat $O/${TAG}_elimination_steps.txt | head -100; tail -3 $T/log.txt
rm -rf $T
