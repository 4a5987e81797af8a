#!/bin/bash
mkdir -p gpurun_out/r2q
O=gpurun_out/r2q
timeout -k 10 600 python -m pytest tests/test_gpu_towers.py tests/test_gpu_models.py tests/test_gpu_graphstep.py tests/test_gpu_inference.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest.log
