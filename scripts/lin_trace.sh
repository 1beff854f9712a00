#!/bin/bash
# rocprofv3 kernel trace of the parametric residual / Jacobian kernels (k_lin<kind>) at the sizes of BASELINE configs[1] and configs[4]
#   scripts/lin_trace.sh [tag] -> gpurun_out/<tag>_linearize_trace.md
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=/tmp/lin_trace; TAG=${1:-r05}; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
cat > $T/run.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, rome_jl_amd as R
from rome_jl_amd.parametric import _Problem
from rome_jl_amd.distributed import LinearizeShard
dev = torch.device("cuda", 0)
for fg in (R.dead_reckon_init_pose3(R.synth_helix3d(P=10000, N=8), seed=7), R.dead_reckon_init(R.loadG2o("$R/tests/golden/manhattan.g2o", N=8), seed=1),
           R.dead_reckon_init(R.synth_mit_br(P=808, n_landmarks=120, N=8), seed=1)):
    P = _Problem(fg)
    X = P.pack(R.initParametric(fg, refine=False))
    sh = LinearizeShard(torch, None, 1, 0, device=dev)
    for it in range(50):
        P.linearize(X, None, sh)
    print({k: len(g["mu"]) for k, g in P.groups.items()}, sh.stats)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $T/out -o g -- python $T/run.py > $T/log.txt 2>&1
tail -4 $T/log.txt
python3 $R/scripts/rocpd_summary.py $(find $T/out -name "*_results.db" | head -1) $O/${TAG}_linearize_trace.md | grep -i "k_lin\|kernel |"
rm -rf $T
