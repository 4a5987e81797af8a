#!/bin/bash
mkdir -p gpurun_out/r2t
timeout -k 10 400 python -m pytest tests/test_gpu_models.py -x -q -s -k "fusion_vs_oracle" -p no:cacheprovider > gpurun_out/r2t/pytest.log 2>&1; echo "rc=$?"
grep "fusion vs oracle\|passed\|failed\|Error" gpurun_out/r2t/pytest.log | cut -c1-1500
