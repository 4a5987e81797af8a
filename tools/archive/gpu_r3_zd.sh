#!/bin/bash
# round 3: long transforms on 16 waves (one item per thread and pass) against 8 waves x 4 items — tests, A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3zd; mkdir -p $O
APPLECIDER_FFT_LONG_WIDE=1 timeout -k 10 600 python -m pytest tests/test_gpu_fftconv.py -x -q -p no:cacheprovider > $O/pytest_fft.log 2>&1; rc=$?; tail -3 $O/pytest_fft.log
[ $rc -eq 0 ] || exit $rc
APPLECIDER_FFT_LONG_WIDE=1 timeout -k 10 200 python tools/bench_fftconv.py > $O/fftconv_wide.txt 2>&1; head -3 $O/fftconv_wide.txt
for T in 0 1 0 1; do
if [ $T -eq 1 ]; then export APPLECIDER_FFT_LONG_WIDE=1; else unset APPLECIDER_FFT_LONG_WIDE; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fast-mode --no-ceilings --no-graph > $O/bench_w$T.json 2> $O/bench_w$T.err; echo "wide $T: $(cut -c100-130 $O/bench_w$T.json) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_w$T.json | head -1)"
done
