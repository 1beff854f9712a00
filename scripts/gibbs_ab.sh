#!/bin/bash
# scripts/gibbs_ab.sh lib1.so lib2.so ... : k_product_gibbs / k_gibbs_trees / k_kde_bandwidth_fast average duration (rocprofv3 kernel trace of six
# solve iterations on the Manhattan graph) for each library variant
R=${GRAFT_REPO_ROOT:-/root/repo}; T=/tmp/gibbs_ab; mkdir -p $T
cd /tmp && export TMPDIR=/tmp
cat > $T/run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, rome_jl_amd as R
fg = R.loadG2o("$R/tests/golden/manhattan.g2o", N=100); R.dead_reckon_init(fg, seed=1)
dg = R.DeviceGraph(fg); dg.upload_beliefs(fg)
o = R.make_opts(N=100, solver=1, seed=11)
for s in range(8):
    dg.conv_step(o, s); dg.product_step(o, s, "lcv", "gibbs")
torch.cuda.synchronize()
PY
for so in "$@"; do
  rm -rf $T/out
  ROME_MI355_LIB=$R/$so timeout 300 rocprofv3 --kernel-trace --stats -d $T/out -o g -- python $T/run.py > $T/log.txt 2>&1
  echo "== $so"
  python3 $R/scripts/rocpd_summary.py $(find $T/out -name "*_results.db" | head -1) | grep -E "k_product_gibbs|k_gibbs_trees|k_kde_bandwidth" | awk -F'|' '{print $2, "avg us", $5, "min", $6}' | cut -c1-160
done
rm -rf $T
