#!/bin/bash
mkdir -p gpurun_out/r2y
timeout -k 10 300 python tools/exp_cu_partition.py bf16x3 > gpurun_out/r2y/part.txt 2>&1; echo rc=$?
grep -v amdgpu gpurun_out/r2y/part.txt
