#!/bin/bash
# gemm_x3 loop work: correctness of everything that multiplies, then the per-K-tile microbench, the TN sweep and a bench line.
# usage: gpurun -- "bash tools/gpu_x3.sh r4x"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-x3}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_ops.py tests/test_gpu_tail.py tests/test_gpu_attention.py tests/test_gpu_fftconv.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/summary.txt; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/bench_x3_ktile.py > $O/x3_ktile.txt 2>&1; echo "ktile rc=$?" | tee -a $O/summary.txt
grep -v amdgpu $O/x3_ktile.txt
timeout -k 10 200 python tools/bench_tail.py > $O/tail.txt 2>&1; echo "tail rc=$?" | tee -a $O/summary.txt
grep -v amdgpu $O/tail.txt | tail -12
timeout -k 10 200 python tools/bench_tn_x3.py > $O/tn_sweep.txt 2>&1; echo "tn rc=$?" | tee -a $O/summary.txt
grep -v amdgpu $O/tn_sweep.txt | cut -c1-150
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --no-ceilings --no-graph > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
l=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms", l["ms_per_step"], "roofline", l["roofline"]["kernel"], l["roofline"]["frac"], l["roofline"]["ms_per_step"])
PY
