#!/bin/bash
# kernel trace of one solve iteration with the reference's product (manikde! bandwidths + multiscale Gibbs product) on the Manhattan graph
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=/tmp/gibbs_trace; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
cat > $T/run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, rome_jl_amd as R
fg = R.loadG2o("$R/tests/golden/manhattan.g2o", N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
for s in range(6):
    dg.conv_step(o, s); dg.product_step(o, s, "lcv", "gibbs")
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $T/out -o g -- python $T/run.py > $T/log.txt 2>&1
python3 $R/scripts/rocpd_summary.py $(find $T/out -name "*_results.db" | head -1) $O/${1:-r02}_solve_iteration_trace.md
rm -rf $T
