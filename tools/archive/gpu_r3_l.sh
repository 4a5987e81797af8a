#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_fftconv.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/summary.txt
tail -25 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_fftconv.py f32 > $O/bench_f32.txt 2>&1; echo "bench f32 rc=$?" | tee -a $O/summary.txt
cat $O/bench_f32.txt
timeout -k 10 300 python tools/bench_fftconv.py bf16x3 > $O/bench_x3.txt 2>&1; echo "bench x3 rc=$?" | tee -a $O/summary.txt
cat $O/bench_x3.txt
