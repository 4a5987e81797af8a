"""Skinny-K NT products (encoder / ConvNeXt MLP shapes): cost of the epilogue variants."""
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

def run(M, N, K):
    a16, b16 = H.cast16(torch.randn(M, K, device=dev)), H.cast16(torch.randn(N, K, device=dev))
    c = torch.empty(M, N, device=dev); pre = torch.empty(M, N, device=dev)
    c16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    am, bm = H.mat(H._p(a16), K), H.mat(H._p(b16), K)
    v = {
        "f32 out": lambda: H.gemm(0, M, N, K, am, bm, H.mat(H._p(c), N), math=2),
        "f32+bias": lambda: H.gemm(0, M, N, K, am, bm, H.mat(H._p(c), N), math=2, bias=bias),
        "c16 only": lambda: H.gemm(0, M, N, K, am, bm, H.mat(None, N), math=2, c16=c16, ld_c16=N),
        "c16 relu": lambda: H.gemm(0, M, N, K, am, bm, H.mat(None, N), math=2, c16=c16, ld_c16=N, bias=bias, act=2),
        "c16 relu drop": lambda: H.gemm(0, M, N, K, am, bm, H.mat(None, N), math=2, c16=c16, ld_c16=N, bias=bias, act=2, drop_p=0.4, drop_seed=123),
        "c16 gelu pre": lambda: H.gemm(0, M, N, K, am, bm, H.mat(None, N), math=2, c16=c16, ld_c16=N, bias=bias, act=1, pre_out=pre, ld_pre=N),
        "tile2 f32": lambda: H.gemm(0, M, N, K, am, bm, H.mat(H._p(c), N), math=2, tile=2),
        "tile3 f32": lambda: H.gemm(0, M, N, K, am, bm, H.mat(H._p(c), N), math=2, tile=3),
    }
    out = [f"{k}: {timeit(f):.1f}" for k, f in v.items()]
    mb = (M * K * 2 + M * N * 4) / 1e6
    print(f"NT M{M} N{N} K{K} ({mb:.0f} MB f32-out): " + " | ".join(out), flush=True)

run(66048, 512, 128)
run(115200, 384, 96)
run(66048, 128, 512)
run(4608, 1536, 384)
run(4608, 384, 1536)
run(2097152, 192, 64)
