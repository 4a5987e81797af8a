"""Split-K sweep for the skinny weight-gradient (TN) shapes (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

def tn(M, N, K, splits):
    a16, b16 = H.cast16(torch.randn(K, M, device=dev)), H.cast16(torch.randn(K, N, device=dev))
    c = torch.zeros(M, N, device=dev)
    out = []
    for sp in splits:
        f = lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a16), M), H.mat(H._p(b16), N), H.mat(H._p(c), N), math=2, accumulate=2, split_k=sp)
        out.append(f"{sp}:{timeit(f):.1f}us")
    print(f"TN M{M} N{N} K{K} (auto {H._split_for(M, N, K)}): " + "  ".join(out), flush=True)

tn(512, 128, 66048, [4, 8, 16, 32, 64, 129, 256])
tn(128, 512, 66048, [4, 8, 16, 32, 64, 129])
tn(128, 128, 66048, [8, 16, 32, 64, 129, 256])
tn(384, 1536, 4608, [1, 2, 4, 9, 18])
tn(96, 384, 115200, [8, 16, 32, 64, 128, 225])
tn(192, 768, 25088, [4, 8, 16, 32, 49, 98])
tn(64, 192, 2097152, [32, 64, 128, 256, 512])
tn(128, 384, 524288, [32, 64, 128, 341])
tn(768, 3072, 512, [1, 2, 4])
