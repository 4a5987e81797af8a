#!/bin/bash
# round 3 (frequency-domain convs): default bench line + single-stream kernel statistics + per-branch times
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams --no-h2d --no-ceilings --no-graph > $R/$O/prof.log 2>&1); echo "stats rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/kernel_stats_single_stream.csv
rm -rf $O/prof
head -45 $O/kernel_stats_single_stream.csv | cut -c1-200
timeout -k 10 200 python tools/exp_branch_times.py bf16x3 > $O/branch_times.txt 2>&1; echo "branches rc=$?" | tee -a $O/summary.txt
cat $O/branch_times.txt
