#!/bin/bash
# pass M: device step counter + whole-step hipGraph tests, bench with the hip_graph leg
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
timeout -k 10 500 python -m pytest tests/test_gpu_graphstep.py -x -q > $O/pytest_graph.log 2>&1; echo "pytest graph rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_graph.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r2m/bench_x3.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "graph", d.get("hip_graph"), "fast", {k: d["fast_mode"][k] for k in ("value", "ms_per_step", "hip_graph")})
PY
