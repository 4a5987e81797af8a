#!/bin/bash
mkdir -p gpurun_out/r2w
timeout -k 10 500 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r2w/soak.json 2> gpurun_out/r2w/soak.err; echo "soak rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r2w/soak.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["final_loss"], d["hip_graph"]["ms_per_step"], d["fast_mode"]["ms_per_step"])
PY
