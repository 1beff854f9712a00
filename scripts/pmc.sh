#!/bin/bash
# HBM traffic counters (separate --pmc passes, kernel-trace only; MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/cal_$c -o cal -- $R/scripts/ubench/copy8 > $O/cal_$c.log 2>&1
  for s in newton closed_form; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/${s}_$c -o $s -- python $R/bench.py --solver $s --steps 20 --warmup 2 --no-cpu-baseline --no-modes > $O/${s}_$c.log 2>&1
  done
done
ls -R $O | head -30
