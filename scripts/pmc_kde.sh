#!/bin/bash
# SQ counters of the KDE kernels (scripts/kde_profile.py) -- separate --pmc pass, kernel-trace only
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_kde; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/a -o a -- python $R/scripts/kde_profile.py > $O/a.log 2>&1
python3 - <<PY
import sqlite3,glob
for d in sorted(glob.glob('$O/a/*_results.db')):
    db=sqlite3.connect(d)
    for r in db.execute("select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection group by kernel_name, counter_name"):
        if 'kde' in r[0]: print(r[0][:40], r[1], r[2], '%.5g'%r[3], '%.5g'%r[4])
    for r in db.execute("select name, count(*), avg(duration), max(duration) from kernels group by name"):
        if 'kde' in r[0]: print('duration_ns', r[0][:40], r[1], r[2], r[3])
PY
