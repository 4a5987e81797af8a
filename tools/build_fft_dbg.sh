#!/bin/bash
# Diagnostic builds of the transform kernels for the co-residency question of DESIGN section 7-9 (not part of the
# product; the probes that load them are tools/archive/dbg_coresidency6.py / 7.py).  The instrumented source is the
# product's ac_fft.hip + tools/fft_debug.patch (in-kernel self-checks, barrier variants), applied to a temporary copy.
#   usage: tools/build_fft_dbg.sh <variant>      -> tools/libac_dbg_<variant>.so
#     checks  -DAC_FFT_DEBUG: self-checks of every load / LDS hand-over / barrier; packed fp32 ON (the checks only mean
#             something while the fault is present)
#     sync1   two barriers in a row at every phase boundary, packed fp32 ON
#     sync2   barrier + s_sleep, packed fp32 ON
#     nopk    the instrumented source with the product's flags (no packed fp32): the fault is gone
# The diagnostic probes ask for the exact LDS size (ac_fft_rows_desc.lds_exact = 1) so that workgroups share their CU.
set -e
V=${1:?variant: checks | sync1 | sync2 | nopk}
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/applecider_amd/csrc
T=$(mktemp -d)
mkdir -p $T/applecider_amd/csrc $T/include
cp $C/ac_fft.hip $C/ac_fft_core.h $C/ac_common.h $T/applecider_amd/csrc/
cp $R/include/applecider_hip.h $T/include/
(cd $T && patch -s -p1 < $R/tools/fft_debug.patch)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize"
case $V in
  checks) FLAGS="$FLAGS -DAC_FFT_DEBUG" ;;
  sync1)  FLAGS="$FLAGS -DAC_FFT_SYNC_VARIANT=1" ;;
  sync2)  FLAGS="$FLAGS -DAC_FFT_SYNC_VARIANT=2" ;;
  nopk)   FLAGS="$FLAGS -DAC_FFT_DEBUG -Xclang -target-feature -Xclang -packed-fp32-ops" ;;
  *) echo "unknown variant $V"; exit 2 ;;
esac
OBJS=$(ls $C/*.o | grep -v "/f16_" | grep -v "/ac_fft.o")
/opt/rocm/bin/hipcc $FLAGS -c $T/applecider_amd/csrc/ac_fft.hip -o $T/ac_fft_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/libac_dbg_$V.so $OBJS $T/ac_fft_dbg.o
rm -rf $T
ls -la $R/tools/libac_dbg_$V.so
