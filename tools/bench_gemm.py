"""Micro-benchmark of the gather-GEMM kernels (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def plain(M, N, K, math):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    if math == 2:
        a16, b16 = H.cast16(a), H.cast16(b)
        f = lambda: H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a16), K), H.mat(H._p(b16), K), H.mat(H._p(c), N), math=2)
    else:
        f = lambda: H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c), N), math=math)
    ms = timeit(f)
    print(f"plain NT M{M} N{N} K{K} math{math}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TF")

def conv(B, L, Cin, Cout, k, math):
    P = k // 2; Lp = L + 2 * P
    xpad = torch.randn(B, Lp, Cin, device=dev); w = torch.randn(Cout, k * Cin, device=dev); y = torch.empty(B, L, Cout, device=dev)
    if math == 2:
        x16, w16 = H.cast16(xpad), H.cast16(w)
        f = lambda: H.gemm(H.AC_GEMM_NT, B * L, Cout, k * Cin, H.mat(H._p(x16), r1=L, r2=L, s1=Lp * Cin, s3=Cin), H.mat(H._p(w16), k * Cin), H.mat(H._p(y), Cout), math=2)
    else:
        f = lambda: H.gemm(H.AC_GEMM_NT, B * L, Cout, k * Cin, H.mat(H._p(xpad), r1=L, r2=L, s1=Lp * Cin, s3=Cin), H.mat(H._p(w), k * Cin), H.mat(H._p(y), Cout), math=math)
    ms = timeit(f, 3)
    print(f"conv fwd B{B} L{L} Cin{Cin} Cout{Cout} k{k} math{math}: {ms:.3f} ms  {2*B*L*Cout*k*Cin/ms/1e9:.1f} TF")

def tn(M, N, K, math, split):
    a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev); c = torch.zeros(M, N, device=dev)
    if math == 2:
        a16, b16 = H.cast16(a), H.cast16(b)
        f = lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a16), M), H.mat(H._p(b16), N), H.mat(H._p(c), N), math=2, accumulate=2, split_k=split)
    else:
        f = lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a), M), H.mat(H._p(b), N), H.mat(H._p(c), N), math=math, accumulate=2, split_k=split)
    ms = timeit(f, 3)
    print(f"TN M{M} N{N} K{K} math{math} split{split}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TF")


def conv_t(B, L, Cin, Cout, k, tile):
    P = k // 2; Lp = L + 2 * P
    xpad = torch.randn(B, Lp, Cin, device=dev); w = torch.randn(Cout, k * Cin, device=dev); y = torch.empty(B, L, Cout, device=dev)
    x16, w16 = H.cast16(xpad), H.cast16(w)
    f = lambda: H.gemm(H.AC_GEMM_NT, B * L, Cout, k * Cin, H.mat(H._p(x16), r1=L, r2=L, s1=Lp * Cin, s3=Cin), H.mat(H._p(w16), k * Cin), H.mat(H._p(y), Cout), math=2, tile=tile)
    ms = timeit(f, 3)
    print(f"conv fwd B{B} L{L} Cin{Cin} Cout{Cout} k{k} tile{tile}: {ms:.3f} ms  {2*B*L*Cout*k*Cin/ms/1e9:.1f} TF")

def plain_t(M, N, K, tile, mode="NT", split=1):
    c = torch.zeros(M, N, device=dev)
    if mode == "NT":
        a16, b16 = H.cast16(torch.randn(M, K, device=dev)), H.cast16(torch.randn(N, K, device=dev))
        f = lambda: H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a16), K), H.mat(H._p(b16), K), H.mat(H._p(c), N), math=2, tile=tile)
    else:
        a16, b16 = H.cast16(torch.randn(K, M, device=dev)), H.cast16(torch.randn(K, N, device=dev))
        f = lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a16), M), H.mat(H._p(b16), N), H.mat(H._p(c), N), math=2, tile=tile, accumulate=2, split_k=split)
    ms = timeit(f, 3)
    print(f"{mode} M{M} N{N} K{K} tile{tile} split{split}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TF")


def win(B, L, C, N, k, flip):
    P = k // 2; Lp = L + 2 * P
    a = H.cast16(torch.randn(B, Lp, C, device=dev)); w = H.cast16(torch.randn(N, k * C, device=dev)); y = torch.empty(B, L, N, device=dev)
    for on in (True, False):
        if on:
            f = lambda: H.conv_window(a, Lp * C, C, 0, 0, B, L, C, k, w, k * C, C, flip, N, H._p(y), N, None, False)
            assert f()
        else:
            f = lambda: H.gemm(H.AC_GEMM_NT, B * L, N, k * C, H.mat(H._p(a), r1=L, r2=L, s1=Lp * C, s3=C), H.mat(H._p(w), k * C), H.mat(H._p(y), N), math=2)
        ms = timeit(f, 3)
        print(f"{'window' if on else 'generic'} B{B} L{L} C{C} N{N} k{k}: {ms:.3f} ms  {2*B*L*N*k*C/ms/1e9:.1f} TF")

for variant in (1, 0):
    H._CONVWIN_VARIANT = variant
    print("window variant", variant, "(1 = 4 waves, 0 = auto: two K-parity groups for N <= 64)")
    win(512, 1024, 128, 64, 251, True)    # stage-2 dX, k = 251
    win(512, 1024, 128, 64, 31, True)
    win(512, 256, 256, 128, 61, True)     # stage-3 dX: 128-row tiles (the 256-row window does not fit)
    win(512, 256, 256, 128, 15, True)
H._CONVWIN_VARIANT = 0
win(512, 1024, 64, 128, 251, False)   # stage-2 fwd
