#!/bin/bash
# quick pass: GPU suite (optionally a -k filter as $2) + the default bench line.  usage: gpurun -- "bash tools/gpu_quick.sh r4a [-k expr]"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-quick}; mkdir -p $O
shift
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider "$@" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/summary.txt
tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
cut -c1-400 $O/bench_default.json
