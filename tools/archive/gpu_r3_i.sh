#!/bin/bash
# round 3, pass I: fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the bf16x3 step at HEAD; inference
# and parity reports; default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
R=$GRAFT_REPO_ROOT
MODE=bf16x3
for CTR in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR -d $R/$O/pmc_$CTR -- python3 $R/bench.py --steps 1 --warmup 1 --math $MODE --no-cpu-baseline --no-fast-mode --no-graph --no-branch-streams --no-h2d --no-ceilings > $R/$O/pmc_$CTR.log 2>&1); echo "pmc $CTR rc=$?" | tee -a $O/summary.txt
  DB=$(find $O/pmc_$CTR -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py $DB $CTR > $O/pmc_$CTR.json
  rm -rf $O/pmc_$CTR
done
python tools/pmc_merge.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json > $O/r03_pmc_hbm_traffic_$MODE.json
grep -E "window|wgrad|gemm_x3|dwconv_pipe" -A4 $O/r03_pmc_hbm_traffic_$MODE.json | head -70
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
cut -c1-300 $O/bench_default.json
timeout -k 10 200 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1; echo "shapes rc=$?" | tee -a $O/summary.txt
head -30 $O/shapes_x3.txt
cat $O/summary.txt
