#!/bin/bash
# gemm_x3 prologue work: product tests, per-shape launch times of the step, bench line.  usage: gpurun -- "bash tools/gpu_x3b.sh r4y2"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-x3b}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_ops.py tests/test_gpu_fftconv.py tests/test_gpu_models.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1; echo "shapes rc=$?"
grep -v amdgpu $O/shapes_x3.txt | head -24
timeout -k 10 400 python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --no-ceilings > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "import json;l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(l['value'],l['ms_per_step'],l['hip_graph']['ms_per_step']);print({k:(v.get('ms_per_step'),(v.get('hbm_bound') or {}).get('GBps')) for k,v in l['roofline']['all_gemm'].items() if k.startswith('gemm')})"
