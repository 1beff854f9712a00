#!/bin/bash
# scripts/solve_counters.sh <tag>   (on the GPU box, through gpurun)
# SQ counters of the solve-iteration kernels (manikde! bandwidths, ball trees, multiscale Gibbs product) on the Manhattan graph:
# VALU instructions per wave and VALU-busy fraction per kernel -> gpurun_out/<tag>_solve_sq_counters.json.  Counters in their own
# passes with --kernel-trace only; raw rocprof output stays in /tmp.
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=/tmp/solve_ctr; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
cat > $T/run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, rome_jl_amd as R
fg = R.loadG2o("$R/tests/golden/manhattan.g2o", N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
for s in range(4):
    dg.conv_step(o, s); dg.product_step(o, s, "lcv", "gibbs")
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $T/a -o a -- python $T/run.py > $T/a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS --kernel-trace -d $T/b -o b -- python $T/run.py > $T/b.log 2>&1
python3 - <<PY
import sqlite3, glob, json
T, O, tag = "$T", "$O", "$tag"
res = {}
for sub in ("a", "b"):
    for d in sorted(glob.glob("%s/%s/**/*_results.db" % (T, sub), recursive=True)):
        db = sqlite3.connect(d)
        try:
            rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
        except Exception as e:
            rows = []
        for name, cn, n, avg in rows:
            for k in ("k_kde_bandwidth_fast", "k_product_gibbs", "k_gibbs_trees"):
                if k in name: res.setdefault(k, {"kernel": name})[cn] = avg; res[k]["launches_averaged"] = n
        for name, avg in db.execute("select name, avg(duration) from kernels group by name"):
            for k in ("k_kde_bandwidth_fast", "k_product_gibbs", "k_gibbs_trees"):
                if k in name: res.setdefault(k, {"kernel": name}).setdefault("_dur", []).append(avg)
for k, r in res.items():
    dur = sum(r.pop("_dur")) / 2.0 if len(r.get("_dur", [])) == 2 else sum(r.pop("_dur", [0])) 
    r["kernel_ns_under_counters"] = dur
    if "SQ_WAVES" in r and "SQ_INSTS_VALU" in r:
        r["derived"] = {"clock_GHz": 2.4, "valu_instructions_per_wave": r["SQ_INSTS_VALU"] / r["SQ_WAVES"],
                        "valu_busy_fraction": 4.0 * r["SQ_ACTIVE_INST_VALU"] / (256 * 4 * 2.4 * dur) if dur else None}
json.dump({"source": "scripts/solve_counters.sh: rocprofv3 --pmc SQ_* --kernel-trace over 4 solve iterations (conv sweep, manikde! bandwidths, "
                     "multiscale Gibbs product) of the Manhattan-3500 graph, N = 100; averages per launch", "kernels": res},
          open("%s/%s_solve_sq_counters.json" % (O, tag), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
tail -3 $T/a.log; rm -rf $T
