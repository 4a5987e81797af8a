#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/bench_dwconv.py > $GRAFT_REPO_ROOT/$O/dw.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB 1 > $O/dw_stats.csv; rm -rf $O/prof
grep dwconv $O/dw_stats.csv | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["fast_mode"]["value"], d.get("roofline_hbm"))
PY
