#!/bin/bash
# diagnostic builds of ac_fft.hip linked with the product objects -> tools/libac_dbg_*.so (not part of the product)
set -e
C=/root/repo/applecider_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize"
OBJS=$(ls $C/*.o | grep -v "/f16_" | grep -v "/ac_fft.o")
/opt/rocm/bin/hipcc $FLAGS -DAC_FFT_DEBUG -c $C/ac_fft.hip -o /tmp/ac_fft_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/libac_dbg_checks.so $OBJS /tmp/ac_fft_dbg.o
ls -la /root/repo/tools/libac_dbg_checks.so
