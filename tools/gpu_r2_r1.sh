#!/bin/bash
# pass R1: full GPU suite, smoke, default bench (+ --h2d, cpu baseline, graph child), 2-rank rehearsal
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python bench.py --h2d > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --force-branch-streams --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_rehearse.json 2> $O/bench_rehearse.err; echo "rehearse rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r2r/bench_default.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "graph", d.get("hip_graph"), "\nfast", d.get("fast_mode"), "\ncpu", d.get("cpu_baseline"), "\nh2d", d.get("pcie_inclusive"), "\nroof", {k: v for k, v in d["roofline"].items() if k != "all_gemm"})
r = json.loads(open('gpurun_out/r2r/bench_rehearse.json').read().strip().splitlines()[-1])
print("rehearse", r["value"], r["n_gpus"], r["rccl_ranks"])
PY
