"""Does the matrix-core attention kernel (the strongest 'culprit' of DESIGN 7-9) disturb the LDS of a co-resident
workgroup of ANOTHER kernel?  tools/lds_probe.hip's pattern kernel (fill the whole dynamic allocation, spin, verify,
rewrite) with the LDS sizes / workgroup sizes of the affected transform kernels, beside attention on a second stream."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H
lib = ctypes.CDLL("tools/liblds_probe.so")
lib.lds_pattern.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
H.set_math("bf16x3")
qkv = torch.randn(512, 129, 384, device=dev); pad = torch.zeros(512, 129, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
for lds, thr in ((27904, 512), (37120, 512), (21000, 512), (110848, 1024), (65536, 256)):
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(30):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.no_grad():
                for _ in range(4):
                    H.mha(qkv, pad, 8, 0.0, False)
        lib.lds_pattern(err.data_ptr(), 2, 4096, thr, lds, 6, torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(f"pattern kernel {lds:6d} B of LDS x {thr:4d} threads, 4096 workgroups, beside attention, 30 runs: {int(err[0])} wrong words", flush=True)
