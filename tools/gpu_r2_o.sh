#!/bin/bash
mkdir -p gpurun_out/r2o
timeout -k 10 300 python -m pytest tests/test_gpu_parity_modes.py -x -q -p no:cacheprovider -k "conv_bank" > gpurun_out/r2o/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2o/pytest.log
