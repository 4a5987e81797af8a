#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
R=$GRAFT_REPO_ROOT
for V in 0 1; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/dw$V -- python3 $R/tools/bench_dwconv.py 512 $V > $R/$O/dw$V.log 2>&1); echo "dw$V rc=$?" | tee -a $O/summary.txt
  DB=$(find $O/dw$V -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 1 > $O/dwconv_v${V}_kernel_stats.csv
  rm -rf $O/dw$V
  grep -i "dwconv" $O/dwconv_v${V}_kernel_stats.csv | cut -c1-170
done
