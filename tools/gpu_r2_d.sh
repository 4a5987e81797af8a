#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_wgrad.py tests/test_gpu_attention.py -q -x -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest_new.log
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/summary.txt
python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1
cut -c1-600 $O/bench_default.json
head -30 $O/shapes_x3.txt
cat $O/summary.txt
