#!/bin/bash
# pass R2: PMC traffic (FETCH_SIZE / WRITE_SIZE in separate runs) and kernel statistics of both modes, inference bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2r2; mkdir -p $O
for MODE in bf16x3 bf16; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --kernel-trace --pmc $CTR -d $GRAFT_REPO_ROOT/$O/pmc_${MODE}_$CTR -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --math $MODE --no-cpu-baseline --no-fast-mode --no-graph --no-branch-streams > $GRAFT_REPO_ROOT/$O/pmc_${MODE}_$CTR.log 2>&1); echo "pmc $MODE $CTR rc=$?" | tee -a $O/summary.txt
    DB=$(find $O/pmc_${MODE}_$CTR -name "*.db" | head -1)
    [ -n "$DB" ] && python tools/rocpd_pmc.py $DB $CTR > $O/pmc_${MODE}_$CTR.json
    rm -rf $O/pmc_${MODE}_$CTR
  done
  python tools/pmc_merge.py $O/pmc_${MODE}_FETCH_SIZE.json $O/pmc_${MODE}_WRITE_SIZE.json > $O/r02_pmc_hbm_traffic_$MODE.json
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$MODE -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --math $MODE --no-cpu-baseline --no-fast-mode --no-graph --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof_$MODE.log 2>&1); echo "stats $MODE rc=$?" | tee -a $O/summary.txt
  DB=$(find $O/prof_$MODE -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/${MODE}_kernel_stats.csv
  rm -rf $O/prof_$MODE
done
timeout -k 10 300 python tools/bench_infer.py > $O/infer.jsonl 2> $O/infer.err; echo "infer rc=$?" | tee -a $O/summary.txt
timeout -k 10 200 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1
grep -E "window|wgrad|gemm_x3" -A4 $O/r02_pmc_hbm_traffic_bf16x3.json | head -60
head -8 $O/bf16x3_kernel_stats.csv | cut -c1-170
head -8 $O/bf16_kernel_stats.csv | cut -c1-170
cat $O/infer.jsonl | cut -c1-300
cat $O/summary.txt
