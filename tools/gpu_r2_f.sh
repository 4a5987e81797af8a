#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_attention.py tests/test_gpu_models.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/summary.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_x3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof_x3.log 2>&1); echo "prof x3 rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof_x3 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/x3_kernel_stats.csv
rm -rf $O/prof_x3
cut -c1-300 $O/bench_default.json
head -30 $O/x3_kernel_stats.csv | cut -c1-160
cat $O/summary.txt
