#!/bin/bash
# round 3: does a doubled barrier (or barrier + sleep) in the transform kernels change the co-residency fault?
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3zf; mkdir -p $O
export APPLECIDER_FFT_SHARED_CU=1
TAG="exact LDS request, product barriers" timeout -k 10 200 python tools/dbg_coresidency6.py > $O/a.txt 2>&1; cat $O/a.txt | grep -v amdgpu.ids
TAG="exact LDS request, two barriers in a row" APPLECIDER_HIP_LIB=$GRAFT_REPO_ROOT/tools/libac_dbg_sync1.so timeout -k 10 200 python tools/dbg_coresidency6.py > $O/b.txt 2>&1; cat $O/b.txt | grep -v amdgpu.ids
TAG="exact LDS request, barrier + s_sleep" APPLECIDER_HIP_LIB=$GRAFT_REPO_ROOT/tools/libac_dbg_sync2.so timeout -k 10 200 python tools/dbg_coresidency6.py > $O/c.txt 2>&1; cat $O/c.txt | grep -v amdgpu.ids
unset APPLECIDER_FFT_SHARED_CU
TAG="whole-CU request (product default)" timeout -k 10 200 python tools/dbg_coresidency6.py > $O/d.txt 2>&1; cat $O/d.txt | grep -v amdgpu.ids
