#!/bin/bash
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_modes.py tests/test_gpu_models.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
timeout -k 10 300 python tools/shape_profile.py 512 bf16x3 > $O/shapes_x3.txt 2>&1; echo "shapes rc=$?"
grep "total" $O/shapes_x3.txt; grep " TN " $O/shapes_x3.txt | awk '{s+=$1} END{print "TN total ms", s}'; grep " TN " $O/shapes_x3.txt | head -12
