#!/bin/bash
# round 3: stage 2's k = 31 as one 1152-point sequence (shares the spectrum of x and the input-gradient transform with k = 251) or as 3 x 384 windows
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3ze; mkdir -p $O
for W in 1.4 1.0 1.4 1.0; do
APPLECIDER_FFT_LONG_WEIGHT=$W timeout -k 10 300 python bench.py --no-cpu-baseline --no-fast-mode --no-ceilings --no-graph > $O/bench_w$W.json 2> $O/bench_w$W.err; echo "weight $W: $(grep -o '"value": [0-9.]*' $O/bench_w$W.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_w$W.json | head -1)"
done
