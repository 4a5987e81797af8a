#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3x; mkdir -p $O
for e in "X=1" "APPLECIDER_FFT_NO_RADIX9=1" "APPLECIDER_FFT_NO_RADIX9=1 APPLECIDER_FFT_NO_SHARE=1"; do
  env $e timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-fast-mode --no-h2d --no-ceilings > $O/b.log 2>&1
  echo "== $e: $(tail -1 $O/b.log | cut -c95-180)"
done
timeout -k 10 300 python tools/bench_fftconv.py 2>&1 | grep -v amdgpu | tee $O/bench_fftconv.txt
