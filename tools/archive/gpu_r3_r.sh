#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3r; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
timeout -k 10 200 python tools/dbg_graph_fwd.py 2>&1 | grep -v amdgpu | tail -3
