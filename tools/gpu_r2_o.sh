#!/bin/bash
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
timeout -k 10 300 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_parity_modes.py tests/test_gpu_models.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
timeout -k 10 300 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1; echo "shapes rc=$?"
grep "WX3 M=   8192\|WX3 M=  32768\|total" $O/shapes_x3.txt | head -24
