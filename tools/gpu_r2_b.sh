#!/bin/bash
# Round-2 GPU pass B: attention + parity-mode tests, then the full suite, x3 bench + profile.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests/test_gpu_attention.py tests/test_gpu_parity_modes.py tests/test_gpu_wgrad.py -q -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_new.log
python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_attention.py --deselect tests/test_gpu_parity_modes.py --deselect tests/test_gpu_wgrad.py > $O/pytest.log 2>&1; echo "pytest rest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
python bench.py --steps 10 --warmup 3 --math bf16x3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench x3 rc=$?" | tee -a $O/summary.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?" | tee -a $O/summary.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_x3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --math bf16x3 --no-cpu-baseline --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof_x3.log 2>&1); echo "prof x3 rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof_x3 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/x3_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof_bf16.log 2>&1); echo "prof bf16 rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof_bf16 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/bf16_kernel_stats.csv
rm -rf $O/prof_x3 $O/prof_bf16
cut -c1-300 $O/bench_x3.json $O/bench_bf16.json
cat $O/summary.txt
