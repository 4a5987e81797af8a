#!/bin/bash
# pass N: full GPU suite after the ABI change (per-call step pointers), smoke, bench
mkdir -p gpurun_out/r2n
O=gpurun_out/r2n
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r2n/bench_x3.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "graph", d.get("hip_graph", {}).get("ms_per_step"), "fast", {k: d["fast_mode"].get(k) for k in ("value", "ms_per_step")}, d["fast_mode"].get("hip_graph", {}).get("ms_per_step"))
PY
