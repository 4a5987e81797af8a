#!/bin/bash
# kernel statistics of one encoder branch alone.  usage: gpurun -- "bash tools/gpu_branch.sh r4b spectra"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-branch}; W=${2:-spectra}; mkdir -p $O
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/tools/branch_profile.py $W 5 > $R/$O/prof_$W.log 2>&1); echo "stats rc=$?" | tee $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 5 > $O/kernel_stats_$W.csv
rm -rf $O/prof
head -60 $O/kernel_stats_$W.csv | cut -c1-200
