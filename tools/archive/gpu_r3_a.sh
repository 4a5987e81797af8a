#!/bin/bash
# round 3, pass A: the whole GPU suite (bf16x3 golden comparisons in report-only mode: every error is recorded)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
export APPLECIDER_PARITY_REPORT_ONLY=1
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee $O/summary.txt
tail -40 $O/pytest.log
