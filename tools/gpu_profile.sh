#!/bin/bash
# per-shape matrix-core launch times + single-stream kernel statistics of the bf16x3 step.  usage: gpurun -- "bash tools/gpu_profile.sh r4c"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-prof}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1; echo "shapes rc=$?" | tee $O/summary.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams --no-h2d --no-ceilings --no-graph > $R/$O/prof.log 2>&1); echo "stats rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 3 > $O/kernel_stats_single_stream.csv
rm -rf $O/prof
head -40 $O/kernel_stats_single_stream.csv | cut -c1-160
