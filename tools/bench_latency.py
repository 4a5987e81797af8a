"""Per-K-tile latency of the GEMM kernels when a single workgroup (or one per CU) runs."""
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
    return s.elapsed_time(e) / n * 1e3  # us
import itertools
for mode, dbg in itertools.product(("NT", "TN"), (0,)):
    for M, N in ((128, 128), (2048, 2048)):
        res = []
        for K in (2048, 4096, 8192):
            c = torch.zeros(M, N, device=dev)
            if mode == "NT":
                a, b = H.cast16(torch.randn(M, K, device=dev)), H.cast16(torch.randn(N, K, device=dev))
                f = lambda: H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c), N), math=2, force_simple=dbg)
            else:
                a, b = H.cast16(torch.randn(K, M, device=dev)), H.cast16(torch.randn(K, N, device=dev))
                f = lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a), M), H.mat(H._p(b), N), H.mat(H._p(c), N), math=2, force_simple=dbg)
            res.append(timeit(f))
        slope = (res[2] - res[0]) / ((8192 - 2048) / 64)
        print(f"{mode} dbg{dbg} {M}x{N}: K=2048/4096/8192 -> {res[0]:.1f} / {res[1]:.1f} / {res[2]:.1f} us ; {slope*1000:.0f} ns per 64-deep K tile")
