#!/bin/bash
# window-conv work: full GPU suite, spectra-branch kernel statistics, default bench line.  usage: gpurun -- "bash tools/gpu_conv.sh r4c"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-conv}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/tools/branch_profile.py spectra 5 > $R/$O/prof.log 2>&1); echo "stats rc=$?"
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 5 > $O/kernel_stats_spectra.csv
rm -rf $O/prof
grep -i "conv1d\|TOTAL" $O/kernel_stats_spectra.csv | cut -c1-170
timeout -k 10 400 python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --no-ceilings --no-graph > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "import json;l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(l['value'],l['ms_per_step'])"
