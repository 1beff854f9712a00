#!/bin/bash
# rocprofv3 kernel trace of the KDE bandwidth / max kernels at Manhattan scale (scripts/kde_profile.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_kde -o kde -- python $R/scripts/kde_profile.py > $O/prof_kde.log 2>&1
cat $O/prof_kde.log | tail -6
for f in $(find $O/prof_kde -name "*kernel_stats.csv" | head -1); do echo $f; head -8 $f; done
