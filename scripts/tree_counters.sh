#!/bin/bash
# scripts/tree_counters.sh <tag>   (on the GPU box, through gpurun)
# SQ counters of the kernels of a Bayes-tree pass (Manhattan-3500, relative messages): per kernel the launches, VALU instructions per wave and
# the VALU-busy fraction of the WHOLE chip over the kernel's duration -> gpurun_out/<tag>_tree_sq_counters.json.  A tree pass is ~190 steps
# of a few launches each; most of them cover a handful of variables, i.e. the chip is empty and the launch pays one block's latency.
# Counters in their own pass with --kernel-trace only; raw rocprof output stays in /tmp.
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=/tmp/tree_ctr; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $T/a -o a -- python $R/scripts/tree_solve_manhattan.py --messages relative --passes 2 > $T/a.log 2>&1
python3 - <<PY
import sqlite3, glob, json
T, O, tag = "$T", "$O", "$tag"
res = {}
for d in sorted(glob.glob("%s/a/**/*_results.db" % T, recursive=True)):
    db = sqlite3.connect(d)
    try:
        rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        rows = []; print("no counters:", e)
    for name, cn, n, avg, tot in rows:
        for k in ("k_kde_bandwidth_fast", "k_product_gibbs", "k_gibbs_trees", "k_conv<", "k_conv_flat", "k_block_ops"):
            if k in name:
                r = res.setdefault(k, {"kernel": name[:90]}); r[cn + "_sum"] = tot; r["launches"] = n
    for name, n, tot in db.execute("select name, count(*), sum(duration) from kernels group by name"):
        for k in ("k_kde_bandwidth_fast", "k_product_gibbs", "k_gibbs_trees", "k_conv<", "k_conv_flat", "k_block_ops"):
            if k in name:
                r = res.setdefault(k, {"kernel": name[:90]}); r["total_ns"] = r.get("total_ns", 0) + tot
for k, r in res.items():
    if "SQ_WAVES_sum" in r and r.get("total_ns"):
        r["derived"] = {"waves_per_launch": r["SQ_WAVES_sum"] / r["launches"], "valu_instructions_per_wave": r["SQ_INSTS_VALU_sum"] / max(r["SQ_WAVES_sum"], 1),
                        "avg_us_per_launch": r["total_ns"] / r["launches"] / 1e3,
                        "chip_valu_busy_fraction": 4.0 * r["SQ_ACTIVE_INST_VALU_sum"] / (256 * 4 * 2.4 * r["total_ns"])}
json.dump({"source": "scripts/tree_counters.sh: rocprofv3 --pmc SQ_* --kernel-trace over the init pass + two Bayes-tree passes (relative messages) of Manhattan-3500, N = 100; "
                     "sums over all launches of a kernel; chip_valu_busy_fraction = VALU-busy SIMD cycles / (1024 SIMDs x kernel time): how full the chip is", "kernels": res},
          open("%s/%s_tree_sq_counters.json" % (O, tag), "w"), indent=1)
for k, r in res.items():
    print(k, r.get("launches"), r.get("derived"))
PY
tail -2 $T/a.log; rm -rf $T
