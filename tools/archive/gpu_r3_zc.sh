#!/bin/bash
# round 3: rows transforms that walk several tiles per workgroup with the next tile's loads in flight — tests, A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3zc; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fftconv.py tests/test_gpu_graphstep.py -x -q -p no:cacheprovider > $O/pytest_fft.log 2>&1; rc=$?; tail -3 $O/pytest_fft.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/bench_fftconv.py > $O/fftconv.txt 2>&1; tail -12 $O/fftconv.txt
for T in 4 1 2 8; do
APPLECIDER_FFT_TILES_PER_WG=$T timeout -k 10 300 python bench.py --no-cpu-baseline --no-fast-mode --no-ceilings --no-graph > $O/bench_t$T.json 2> $O/bench_t$T.err; echo "tiles/wg $T: $(cut -c100-190 $O/bench_t$T.json)"
done
