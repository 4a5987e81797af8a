"""Split-K sweep of the split-bf16 weight-gradient products (TN) at the step's shapes: time, GB/s of algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')
H.set_math("bf16x3")

def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

shapes = [(96, 384, 115200), (384, 96, 115200), (512, 128, 66048), (128, 512, 66048), (384, 128, 66048), (128, 128, 66048),
          (384, 1536, 4608), (192, 768, 25088), (256, 768, 131072), (512, 1536, 32768)]
for M, N, K in shapes:
    a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev); c = torch.zeros(M, N, device=dev)
    cur = H._split_for(M, N, K)
    byts = 4.0 * K * (M + N) + 8.0 * M * N
    out = []
    for split in sorted({cur, 8, 16, 32, 64, 128, 256, 512, 1024}):
        if split > K // 32:
            continue
        ms = timeit(lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a), M), H.mat(H._p(b), N), H.mat(H._p(c), N), accumulate=2, split_k=split))
        out.append(f"{'*' if split == cur else ' '}{split}:{ms * 1e3:.0f}us/{byts / ms / 1e6:.0f}GBs")
    print(f"TN M{M} N{N} K{K}: " + "  ".join(out), flush=True)
