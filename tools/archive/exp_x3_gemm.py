"""Probe: split-bf16 GEMM (math 3) on the shapes that dominate gemm_x3's time.  usage: exp_x3_gemm.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from tools.bench_gemm import timeit
dev = torch.device('cuda')
print("lib", os.environ.get("APPLECIDER_HIP_LIB", "default"))
for (M, N, K) in [(262144, 512, 1028), (4608, 384, 1536), (4608, 1536, 384), (2097152, 64, 192), (115200, 384, 96),
                  (66048, 512, 128), (66048, 128, 512), (524288, 128, 384)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    f = lambda: H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c), N), math=3)
    ms = timeit(f, 5)
    print(f"NT M{M} N{N} K{K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TF  {(M*K+N*K+M*N)*4/ms/1e6:.0f} GB/s")
