#!/bin/bash
# kernel trace of the Bayes-tree solve of Manhattan-3500 (3 passes, relative messages) + per-level wall-clock of both passes
#   scripts/tree_trace.sh [tag]   -> gpurun_out/<tag>_tree_kernel_trace.md, gpurun_out/<tag>_tree_solve.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=/tmp/tree_trace; TAG=${1:-r05}; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
rm -f $O/${TAG}_tree_solve.txt
timeout 600 python $R/scripts/tree_solve_manhattan.py --messages relative --passes 3 --levels --out $O/${TAG}_tree_solve.txt > $T/solve.log 2>&1
timeout 600 python $R/scripts/tree_solve_manhattan.py --messages marginal --passes 2 --out $O/${TAG}_tree_solve.txt >> $T/solve.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $T/out -o g -- python $R/scripts/tree_solve_manhattan.py --messages relative --passes 3 > $T/log.txt 2>&1
python3 $R/scripts/rocpd_summary.py $(find $T/out -name "*_results.db" | head -1) $O/${TAG}_tree_kernel_trace.md > /dev/null
tail -5 $T/log.txt
rm -rf $T
