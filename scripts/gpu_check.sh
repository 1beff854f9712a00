#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprofv3 kernel-trace of the bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log
(timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3) > $O/smoke.log
(timeout 400 python bench.py 2>&1 | tail -2) > $O/bench.log
cd /tmp && export TMPDIR=/tmp
for s in newton closed_form nelder_mead; do
  st=2000; wu=2000; [ $s = nelder_mead ] && st=20 && wu=10
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$s -o $s -- python $R/bench.py --solver $s --steps $st --warmup $wu --no-cpu-baseline --no-modes > $O/prof_$s.log 2>&1
done
cd $R
cat $O/pytest.log $O/smoke.log $O/bench.log
find $O -name "*kernel_stats.csv" | head; for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -5 $f; done
