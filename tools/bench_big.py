"""The three large matrix-core shapes of the step (stage-2 conv dW / generic conv / plain GEMMs)."""
import os, sys
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

def plain(mode, M, N, K, split=1, tile=0):
    c = torch.zeros(M, N, device=dev)
    if mode == "NT":
        a16, b16 = H.cast16(torch.randn(M, K, device=dev)), H.cast16(torch.randn(N, K, device=dev))
        f = lambda: H.gemm(0, M, N, K, H.mat(H._p(a16), K), H.mat(H._p(b16), K), H.mat(H._p(c), N), math=2, tile=tile)
    else:
        a16, b16 = H.cast16(torch.randn(K, M, device=dev)), H.cast16(torch.randn(K, N, device=dev))
        f = lambda: H.gemm(2, M, N, K, H.mat(H._p(a16), M), H.mat(H._p(b16), N), H.mat(H._p(c), N), math=2, accumulate=2, split_k=split)
    ms = timeit(f)
    print(f"{mode} M{M} N{N} K{K} split{split} tile{tile}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TF", flush=True)

def conv_dw(B, L, Cin, Cout, k, split, tile=0):
    P = k // 2; Lp = L + 2 * P
    x16 = H.cast16(torch.randn(B, Lp, Cin, device=dev)); dy16 = H.cast16(torch.randn(B * L, Cout, device=dev))
    dw = torch.zeros(Cout, k * Cin, device=dev)
    f = lambda: H.gemm(2, Cout, k * Cin, B * L, H.mat(H._p(dy16), Cout), H.mat(H._p(x16), r1=L, r2=L, s1=Lp * Cin, s3=Cin),
                       H.mat(H._p(dw), k * Cin), accumulate=2, split_k=split, math=2, tile=tile)
    ms = timeit(f, 3)
    print(f"conv dW B{B} L{L} Cin{Cin} Cout{Cout} k{k} split{split} tile{tile}: {ms:.3f} ms  {2*B*L*Cout*k*Cin/ms/1e9:.1f} TF", flush=True)

def conv_fwd(B, L, Cin, Cout, k, tile=0):
    P = k // 2; Lp = L + 2 * P
    x16 = H.cast16(torch.randn(B, Lp, Cin, device=dev)); w16 = H.cast16(torch.randn(Cout, k * Cin, device=dev))
    y = torch.empty(B * L, Cout, device=dev)
    f = lambda: H.gemm(0, B * L, Cout, k * Cin, H.mat(H._p(x16), r1=L, r2=L, s1=Lp * Cin, s3=Cin), H.mat(H._p(w16), k * Cin), H.mat(H._p(y), Cout), math=2, tile=tile)
    ms = timeit(f, 3)
    print(f"conv fwd (generic NT) B{B} L{L} Cin{Cin} Cout{Cout} k{k} tile{tile}: {ms:.3f} ms  {2*B*L*Cout*k*Cin/ms/1e9:.1f} TF", flush=True)

for t in (0, 3):
    plain("NT", 8192, 8192, 4096, tile=t)
    plain("NT", 262144, 512, 1032, tile=t)
    plain("NT", 32768, 256, 15872, tile=t)
    plain("NT", 32768, 512, 7936, tile=t)
    plain("NT", 8192, 512, 13312, tile=t)
    plain("NT", 8192, 1024, 6656, tile=t)
    conv_fwd(512, 1024, 64, 128, 251, tile=t)
    conv_fwd(512, 256, 128, 256, 61, tile=t)
