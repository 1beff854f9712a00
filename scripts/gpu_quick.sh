#!/bin/bash
# quick GPU check (via gpurun): parity tests, then short bench lines per solver.  scripts/gpu_quick.sh [pytest-args]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q $@ 2>&1 | tail -40) > $O/pytest.log
for s in newton closed_form gauss_newton; do
  (timeout 300 python bench.py --solver $s --no-cpu-baseline --no-modes 2>&1 | tail -1) > $O/bench_$s.log
done
cat $O/pytest.log; for s in newton closed_form gauss_newton; do python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$s.log").read().strip().splitlines()[-1]); print("$s", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_launch"], j["roofline"]["frac"])
except Exception as e:
    print("$s", "ERR", e, open("$O/bench_$s.log").read()[-600:])
PY
done
