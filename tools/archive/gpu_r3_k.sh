#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_models.py -q -p no:cacheprovider -k "conv_bank or spectranet or stage1" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -12 $O/pytest.log
timeout -k 10 200 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1; echo "shapes rc=$?" | tee -a $O/summary.txt
head -16 $O/shapes_x3.txt
