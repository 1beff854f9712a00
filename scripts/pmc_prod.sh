#!/bin/bash
# SQ counters of the product kernel inside the device solve loop (scripts/solve_profile.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_prod; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace -d $O/a -o a -- python $R/scripts/solve_profile.py > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace -d $O/b -o b -- python $R/scripts/solve_profile.py > $O/b.log 2>&1
python3 - <<PY
import sqlite3,glob
for d in sorted(glob.glob('$O/*/*_results.db')):
    db=sqlite3.connect(d)
    for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if 'k_product' in r[0]: print(r[1], r[2], '%.4g'%r[3])
    for r in db.execute("select name, avg(duration) from kernels group by name"):
        if 'k_product' in r[0]: print('duration_ns', r[1])
PY
