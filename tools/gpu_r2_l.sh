#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_attention.py tests/test_gpu_models.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["fast_mode"]["value"], d.get("roofline_hbm"))
PY
head -40 $O/shapes_x3.txt
cat $O/summary.txt
