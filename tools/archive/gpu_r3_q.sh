#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3q; mkdir -p $O
timeout -k 10 300 python tools/dbg_fft_determinism.py 2>&1 | grep -v amdgpu | tee $O/determinism.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -8 $O/pytest.log
