#!/bin/bash
# round 3, pass F: weight-gradient ring kernel: parity + A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_wgrad.py -q -p no:cacheprovider -k "conv_bank or wgrad" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
timeout -k 10 300 python tools/bench_wgrad.py 512 7 4,0 > $O/wgrad_ab.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
cat $O/wgrad_ab.txt
