#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/dw -- python3 $R/tools/probe_dwconv.py > $R/$O/dw.log 2>&1); echo "probe rc=$?" | tee $O/summary.txt
DB=$(find $O/dw -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 1 > $O/probe_stats.csv
rm -rf $O/dw
grep -i "dwconv\|ceil" $O/probe_stats.csv | cut -c1-200
