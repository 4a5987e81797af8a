#!/bin/bash
# depthwise work: tests, kernel times of the shortcut form under the tracer.  usage: gpurun -- "bash tools/gpu_dw.sh r4d"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-dw}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -k "dwconv" -x -q > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/tools/branch_profile.py image 5 > $R/$O/prof.log 2>&1); echo "stats rc=$?"
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 5 > $O/kernel_stats_image.csv
rm -rf $O/prof
grep -i "dwconv\|TOTAL" $O/kernel_stats_image.csv | cut -c1-170
