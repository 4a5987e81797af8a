#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
python -m pytest tests/test_gpu_attention.py tests/test_gpu_models.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/summary.txt
for MODE in bf16x3 bf16; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --kernel-trace --pmc $CTR -d $GRAFT_REPO_ROOT/$O/pmc_${MODE}_$CTR -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --math $MODE --no-cpu-baseline --no-fast-mode --no-branch-streams > $GRAFT_REPO_ROOT/$O/pmc_${MODE}_$CTR.log 2>&1); echo "pmc $MODE $CTR rc=$?" | tee -a $O/summary.txt
    DB=$(find $O/pmc_${MODE}_$CTR -name "*.db" | head -1)
    [ -n "$DB" ] && python tools/rocpd_pmc.py $DB $CTR > $O/pmc_${MODE}_$CTR.json
    rm -rf $O/pmc_${MODE}_$CTR
  done
  python tools/pmc_merge.py $O/pmc_${MODE}_FETCH_SIZE.json $O/pmc_${MODE}_WRITE_SIZE.json > $O/r02_pmc_hbm_traffic_$MODE.json
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --math bf16 --no-cpu-baseline --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof_bf16.log 2>&1)
DB=$(find $O/prof_bf16 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/bf16_kernel_stats.csv
rm -rf $O/prof_bf16
cut -c1-300 $O/bench_default.json
grep -E "window|wgrad|gemm_x3|mha" -A4 $O/r02_pmc_hbm_traffic_bf16x3.json | head -60
grep -E "mha" $O/bf16_kernel_stats.csv | cut -c1-160
cat $O/summary.txt
