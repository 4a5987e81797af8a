#!/bin/bash
# round 3: the transform kernels compiled WITHOUT packed fp32 instructions (v_pk_add/mul/fma_f32), exact LDS request, beside attention
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3zg; mkdir -p $O
export APPLECIDER_FFT_SHARED_CU=1
TAG="exact LDS request, packed fp32 (product build)" timeout -k 10 200 python tools/dbg_coresidency6.py > $O/a.txt 2>&1; grep -v amdgpu.ids $O/a.txt
TAG="exact LDS request, NO packed fp32" APPLECIDER_HIP_LIB=$GRAFT_REPO_ROOT/tools/libac_dbg_nopk.so timeout -k 10 200 python tools/dbg_coresidency6.py > $O/b.txt 2>&1; grep -v amdgpu.ids $O/b.txt
