#!/bin/bash
# diagnostic build of ac_fft.hip (-DAC_FFT_DEBUG: in-kernel self-checks) linked with the product objects ->
# tools/libac_dbg_checks.so (not part of the product; run before tools/dbg_coresidency7.py).  The barrier variants of
# tools/gpu_r3_zf.sh are the same recipe with -DAC_FFT_SYNC_VARIANT=1 / 2 -> libac_dbg_sync1.so / libac_dbg_sync2.so; to
# see the fault itself again, build the product library with `make NOPK=` (packed fp32 back on) and run
# tools/dbg_coresidency6.py with APPLECIDER_FFT_SHARED_CU=1.  NOTE: FLAGS below deliberately leave packed fp32 ON for
# ac_fft.hip — the self-checks only mean something while the fault is present.
set -e
C=/root/repo/applecider_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize"
OBJS=$(ls $C/*.o | grep -v "/f16_" | grep -v "/ac_fft.o")
/opt/rocm/bin/hipcc $FLAGS -DAC_FFT_DEBUG -c $C/ac_fft.hip -o /tmp/ac_fft_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/libac_dbg_checks.so $OBJS /tmp/ac_fft_dbg.o
ls -la /root/repo/tools/libac_dbg_checks.so
