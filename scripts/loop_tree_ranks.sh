#!/bin/bash
# Repeats the sharded-tree-solve self-comparison (N processes on one device, real kernels) to expose ordering holes:
#   scripts/loop_tree_ranks.sh REPS [ENV=VALUE ...]   -> gpurun_out/loop_tree_ranks_<tag>.log   (one line per repetition + a summary)
reps=${1:-10}; shift
tag=plain
for kv in "$@"; do export "$kv"; tag="${kv//[^A-Za-z0-9_=]/_}"; done
mkdir -p gpurun_out
log=gpurun_out/loop_tree_ranks_${tag}.log
: > "$log"
fail=0
for i in $(seq 1 "$reps"); do
  out=$(timeout 900 python -m pytest tests/test_gpu_zz_ranks_one_device.py -q -m gpu -k tree_levels -p no:cacheprovider 2>&1 | tail -1)
  echo "rep $i [$*]: $out" >> "$log"
  case "$out" in *failed*|*error*) fail=$((fail + 1));; esac
done
echo "SUMMARY [$*]: $reps repetitions, $fail with failures" >> "$log"
tail -1 "$log"
