#!/bin/bash
# round 3, last pass: the library without packed-fp32 instructions — full GPU suite, bench, the co-residency stress with the
# transform workgroups sharing their CUs again (exact LDS request), graph / transform tests and bench in that mode
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3zh; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/summary.txt; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt; cut -c1-260 $O/bench_default.json
APPLECIDER_FFT_SHARED_CU=1 TAG="product library (no packed fp32), exact LDS request" timeout -k 10 200 python tools/dbg_coresidency6.py > $O/stress_shared.txt 2>&1; grep -v amdgpu.ids $O/stress_shared.txt | tee -a $O/summary.txt
APPLECIDER_FFT_SHARED_CU=1 timeout -k 10 600 python -m pytest tests/test_gpu_graphstep.py tests/test_gpu_fftconv.py tests/test_gpu_parity_modes.py -x -q -p no:cacheprovider > $O/pytest_shared.log 2>&1; echo "pytest (shared CUs) rc=$? $(tail -1 $O/pytest_shared.log)" | tee -a $O/summary.txt
for M in excl shared excl shared; do
  if [ $M = shared ]; then export APPLECIDER_FFT_SHARED_CU=1; else unset APPLECIDER_FFT_SHARED_CU; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-fast-mode --no-ceilings --no-graph > $O/bench_$M.json 2> $O/bench_$M.err
  echo "transform workgroups $M: $(grep -o '"value": [0-9.]*' $O/bench_$M.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$M.json | head -1)" | tee -a $O/summary.txt
done
unset APPLECIDER_FFT_SHARED_CU
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams --no-h2d --no-ceilings --no-graph > $GRAFT_REPO_ROOT/$O/prof.log 2>&1); echo "stats rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/kernel_stats_single_stream.csv
rm -rf $O/prof
cat $O/summary.txt
