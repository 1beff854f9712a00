#!/bin/bash
# A/B of the segmented Gibbs draw (round 6) on the GPU box: wide product (Manhattan sweep), elimination pass with the narrow launches on kSegs lanes
# per sample (default) and on one lane per sample (ROME_GIBBS_SEG_MAXV=0: same draws, bit for bit), and the round-6 baseline library.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for l in scripts/ubench/lib_r6base.so rome.jl_amd/librome_mi355.so; do
  [ -f $l ] || continue
  echo "== $l"
  ROME_MI355_LIB=$R/$l timeout 300 python scripts/gibbs_time.py 2>&1 | tail -2
  for mv in 512 0; do
    [ $l = scripts/ubench/lib_r6base.so -a $mv = 0 ] && continue
    echo "-- ROME_GIBBS_SEG_MAXV=$mv"
    ROME_GIBBS_SEG_MAXV=$mv ROME_MI355_LIB=$R/$l timeout 600 python scripts/elimination_manhattan.py --passes 4 --structures 1 2>&1 | grep -E "per pass|raw|aligned" | head -4
  done
done
