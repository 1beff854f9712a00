#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
s=${1:-newton}
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace -d $O/a_$s -o a -- python $R/bench.py --solver $s --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $O/a_$s.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace -d $O/b_$s -o b -- python $R/bench.py --solver $s --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $O/b_$s.log 2>&1
python3 - <<PY
import sqlite3,glob
for d in sorted(glob.glob('$O/*_$s/*_results.db')):
    db=sqlite3.connect(d)
    for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if 'rome' in r[0]: print(r[1], r[2], '%.4g'%r[3])
    for r in db.execute("select name, avg(duration) from kernels group by name"):
        if 'rome' in r[0]: print('duration_ns', r[1])
PY
