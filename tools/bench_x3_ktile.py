"""Time per 32-deep K tile of gemm_x3_kernel (split-bf16, fp32 operands): one workgroup, one per CU, two per CU.
The slope over K separates the loop from launch + prologue + epilogue.  usage: bench_x3_ktile.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')
H.set_math("bf16x3")

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

Ks = (1024, 2048, 4096)
for mode in ("NT", "NN", "TN"):
    for M, N in ((128, 128), (2048, 2048), (4096, 2048), (8192, 2048)):
        res = []
        for K in Ks:
            c = torch.zeros(M, N, device=dev)
            if mode == "NT":
                a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
                f = lambda: H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c), N))
            elif mode == "NN":
                a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
                f = lambda: H.gemm(H.AC_GEMM_NN, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), N), H.mat(H._p(c), N))
            else:
                a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
                f = lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a), M), H.mat(H._p(b), N), H.mat(H._p(c), N))
            res.append(timeit(f))
        slope = (res[2] - res[0]) / ((Ks[2] - Ks[0]) / 32)
        wgs = (M // 128) * (N // 128)
        tf = 2.0 * M * N * Ks[2] / res[2] / 1e6
        print(f"{mode} {M}x{N} ({wgs} workgroups): K={Ks} -> {res[0]:.1f} / {res[1]:.1f} / {res[2]:.1f} us ; "
              f"{slope * 1000:.0f} ns per 32-deep K tile ; {tf:.0f} TF (1x) at K={Ks[2]}", flush=True)
