#!/bin/bash
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
timeout -k 10 300 python -m pytest tests/test_gpu_parity_modes.py tests/test_gpu_wgrad.py -x -q -p no:cacheprovider -k "x3 or conv or wgrad" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
timeout -k 10 300 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1; echo "shapes rc=$?"
grep "WGR\|total" $O/shapes_x3.txt | head -16
timeout -k 10 300 python tools/shape_profile.py 512 bf16 > $O/shapes_bf16.txt 2>&1; echo "shapes rc=$?"
grep "WGR\|total" $O/shapes_bf16.txt | head -16
