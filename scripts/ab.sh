#!/bin/bash
# A/B of kernel build variants: scripts/ab.sh <solver> lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
s=$1; shift
for so in "$@"; do
  for rep in 1 2; do
  echo -n "$so $s: "
  ROME_MI355_LIB=$R/$so timeout 300 python bench.py --solver $s --steps 2000 --warmup 2000 --no-cpu-baseline --no-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f us/launch  %.3g conv/s' % (1e3*d['roofline']['kernel_ms_per_launch'], d['value']))"
  done
done
