"""Where does a transform workgroup beside the attention kernel first see a wrong value?  Diagnostic build of
ac_fft.hip (-DAC_FFT_DEBUG, tools/libac_dbg_checks.so via APPLECIDER_HIP_LIB) with in-kernel self-checks:
[0] the LDS image differs from the row just written, [1] two reads of a spectrum-store source differ,
[2] read-back of a pass's own store differs, [3] two reads of a pass's source differ, [4] two global reads of a
row differ, [5] read-back of a spectrum store differs, [6] two reads of a twiddle differ, [7] a wave left a barrier before every wave of its workgroup had arrived."""
import ctypes, os, sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H, _lib
dev = torch.device("cuda:0")
H.set_math("bf16x3")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.ac_fft_debug_read.argtypes = [ctypes.POINTER(ctypes.c_uint * 16), ctypes.c_int]
def counters(reset=True):
    buf = (ctypes.c_uint * 16)()
    assert lib.ac_fft_debug_read(ctypes.byref(buf), int(reset)) == 0
    return list(buf)[:13]
g = torch.Generator().manual_seed(0)
x3 = torch.randn(512, 256, 112, generator=g).to(dev)
qkv = torch.randn(512, 129, 384, device=dev); pad = torch.zeros(512, 129, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
for name, size in (("384 points", (7, 1)), ("512 points", 9)):
    vf = lambda: H.fft_rows_fwd(x3, None, 0, 256 * 112, 112, 0, 512, 256, 112, 0, size)
    ref = vf(); torch.cuda.synchronize()
    c0 = counters()
    print(name, "alone: self-check counters", c0[:8], "input / output bit checksums", [hex(v) for v in c0[8:11]], "twiddles off", c0[11], "stage-to-stage XOR", hex(c0[12]), flush=True)
    import numpy as np
    href = int(np.bitwise_xor.reduce(ref.cpu().numpy().view(np.uint32).ravel()))
    print("   XOR of the reference output's bits on the host:", hex(href), flush=True)
    tot = [0] * 8; nbad = 0; runs = 0; xin = 0; xout = 0; xhost = 0; twoff = 0; chain = 0
    for it in range(30):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.no_grad():
                for _ in range(4):
                    H.mha(qkv, pad, 8, 0.0, False)
        out = vf()
        torch.cuda.synchronize()
        c = counters()
        tot = [a + b for a, b in zip(tot, c[:8])]
        d = int((out != ref).sum()); nbad += d; runs += d > 0
        xin += (c[8] != c0[8]) or (c[9] != c0[9]); xout += c[10] != c0[10]; twoff += c[11]; chain += c[12] != 0
        xhost += int(np.bitwise_xor.reduce(out.cpu().numpy().view(np.uint32).ravel())) != c[10]
    print(f"{name} beside attention, 30 runs: {runs} runs differ, {nbad} wrong output values; self-checks "
          f"[fill {tot[0]}, spectrum double-read {tot[1]}, pass read-back {tot[2]}, pass double-read {tot[3]}, global row double-read {tot[4]}, spectrum store read-back {tot[5]}, twiddle double-read {tot[6]}, barrier left early {tot[7]}]; launches whose INPUT checksum differs from the run alone: {xin}; "
          f"whose STORED-value checksum differs from the run alone: {xout}; whose stored-value checksum differs from the XOR of what the host "
          f"finds in the output: {xhost}; twiddles off their analytic value: {twoff}; launches in which a value written by one stage was NOT the value the next stage read (512 points only): {chain}", flush=True)
