#!/bin/bash
# round 3, pass E: full GPU suite on the new defaults (ring window kernel, LayerNorm-backward planes, split-K small
# grids, pipelined depthwise), default bench line, single-stream kernel statistics
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -8 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
cut -c1-600 $O/bench_default.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams --no-h2d --no-ceilings > $R/$O/prof.log 2>&1); echo "stats rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/r03_bench_bf16x3_kernel_stats_single_stream.csv
rm -rf $O/prof
head -30 $O/r03_bench_bf16x3_kernel_stats_single_stream.csv | cut -c1-150
cat $O/summary.txt
