#!/bin/bash
# scripts/solve_phase_trace.sh : per-kernel averages of the solve iteration in the two regimes bench.py's solve leg visits --
# dead-reckoned beliefs (wide) and beliefs initialised from the parametric solution (tight) -- rocprofv3 kernel trace, Manhattan-3500
R=${GRAFT_REPO_ROOT:-/root/repo}; T=/tmp/solve_phase; mkdir -p $T
cd /tmp && export TMPDIR=/tmp
for mode in dead parametric; do
cat > $T/run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, rome_jl_amd as R
fg = R.loadG2o("$R/tests/golden/manhattan.g2o", N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
if "$mode" == "parametric":
    dg.init_from_means(R.solveGraphParametric(fg))
o = R.make_opts(N=100, solver=1, seed=11)
for s in range(10):
    dg.conv_step(o, s); dg.product_step(o, s, "lcv", "gibbs")
torch.cuda.synchronize()
PY
rm -rf $T/out
timeout 300 rocprofv3 --kernel-trace --stats -d $T/out -o g -- python $T/run.py > $T/log.txt 2>&1
echo "== beliefs: $mode"
python3 $R/scripts/rocpd_summary.py $(find $T/out -name "*_results.db" | head -1) | grep -E "k_product_gibbs|k_gibbs_trees|k_kde_bandwidth|k_conv_flat" | awk -F'|' '{print $2, "calls", $3, "avg us", $5, "min", $6, "max", $7}' | cut -c1-200
done
rm -rf $T
