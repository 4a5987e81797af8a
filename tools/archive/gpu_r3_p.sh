#!/bin/bash
# round 3, evidence pass after the frequency-domain convolutions: full GPU suite, default bench line, single-stream
# kernel statistics, fabric traffic (FETCH_SIZE / WRITE_SIZE in separate passes), per-branch times
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/summary.txt
tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
cut -c1-260 $O/bench_default.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams --no-h2d --no-ceilings --no-graph > $R/$O/prof.log 2>&1); echo "stats rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/kernel_stats_single_stream.csv
rm -rf $O/prof
for CTR in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR -d $R/$O/pmc_$CTR -- python3 $R/bench.py --steps 1 --warmup 1 --math bf16x3 --no-cpu-baseline --no-fast-mode --no-graph --no-branch-streams --no-h2d --no-ceilings > $R/$O/pmc_$CTR.log 2>&1); echo "pmc $CTR rc=$?" | tee -a $O/summary.txt
  DB=$(find $O/pmc_$CTR -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py $DB $CTR > $O/pmc_$CTR.json
  rm -rf $O/pmc_$CTR
done
python tools/pmc_merge.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json > $O/pmc_hbm_traffic_bf16x3.json
timeout -k 10 200 python tools/exp_branch_times.py bf16x3 > $O/branch_times.txt 2>&1; echo "branches rc=$?" | tee -a $O/summary.txt
cat $O/branch_times.txt
cat $O/summary.txt
