cd $GRAFT_REPO_ROOT
for v in "" p3w3 p3w4 p3w3sb p3sb; do
  if [ -z "$v" ]; then unset ROME_MI355_LIB; else export ROME_MI355_LIB=$GRAFT_REPO_ROOT/scripts/ubench/lib_$v.so; fi
  echo "== variant: ${v:-default}"
  timeout 200 python scripts/other_factors.py 2>&1 | grep "Pose3Pose3"
done
