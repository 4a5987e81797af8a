#!/bin/bash
mkdir -p gpurun_out/r2v
timeout -k 10 300 python tools/exp_tn_split.py > gpurun_out/r2v/tn.txt 2>&1; echo rc=$?
grep -v amdgpu gpurun_out/r2v/tn.txt
