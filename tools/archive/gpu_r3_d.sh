#!/bin/bash
# round 3, pass D: ring kernel v2 (interleaved fragment reads, one A address per half): parity + A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity_modes.py -q -p no:cacheprovider -k "conv_bank" > $O/pytest_ring.log 2>&1; echo "pytest ring rc=$?" | tee $O/summary.txt
tail -3 $O/pytest_ring.log
timeout -k 10 300 python tools/bench_convx3.py 512 7 4,0 > $O/convx3_ab.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
cat $O/convx3_ab.txt
