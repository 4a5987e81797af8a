#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -6 $O/pytest.log
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/summary.txt
python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1
python tools/bench_infer.py 2048 f16 > $O/infer_f16.json 2> $O/infer_f16.err; echo "infer f16 rc=$?" | tee -a $O/summary.txt
python tools/bench_infer.py 2048 bf16 > $O/infer_bf16.json 2> $O/infer_bf16.err; echo "infer bf16 rc=$?" | tee -a $O/summary.txt
cut -c1-400 $O/bench_default.json; cat $O/infer_f16.json $O/infer_bf16.json | cut -c1-300
head -24 $O/shapes_x3.txt
cat $O/summary.txt
