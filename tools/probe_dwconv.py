"""Where the depthwise forward's time goes: the real kernel, its memory side alone (variant 8), its taps without stores
(variant 9), the round-2 kernel, and a plain 16-byte copy of the same bytes — run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device("cuda")
B, S, C = 512, 15, 96
x = torch.randn(B, S, S, C, device=dev)
w = torch.randn(49, C, device=dev)
b = torch.randn(C, device=dev)
y = torch.empty_like(x)
lib = H._lib_()
for v in (0, 7, 1, 8, 9):
    for _ in range(20):
        H._lib.check(lib.ac_dwconv7x7_fwd_v(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, S, S, C, v, H._stream()), "fwd_v")
    torch.cuda.synchronize()
n = x.numel() * 4
for _ in range(20):
    H._lib.check(lib.ac_ceil_copy(x.data_ptr(), y.data_ptr(), n, H._stream()), "copy")
torch.cuda.synchronize()
print("done", n)
