#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-graph --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB 8 > $O/x3_kernel_stats.csv; rm -rf $O/prof
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["fast_mode"]["value"], d.get("roofline_hbm"), d.get("hip_graph",{}).get("ms_per_step"))
print({k:d["roofline"][k] for k in d["roofline"] if k!="all_gemm"})
PY
head -45 $O/x3_kernel_stats.csv | cut -c1-150
