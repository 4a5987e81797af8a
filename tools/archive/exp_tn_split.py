"""Probe: split-K factor of the long TN products in split-bf16 mode.  usage: exp_tn_split.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from tools.bench_gemm import timeit
dev = torch.device('cuda')
for (M, N, K, splits) in [(512, 1028, 262144, (14, 28, 56, 112)), (64, 192, 2097152, (256, 512, 1024)),
                          (1024, 1536, 8192, (1, 2, 4, 8)), (384, 1536, 4608, (3, 7, 14)), (128, 384, 524288, (64, 128, 256)),
                          (512, 128, 66048, (32, 64, 128))]:
    a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev); c = torch.zeros(M, N, device=dev)
    for sp in splits:
        f = lambda: H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(a), M), H.mat(H._p(b), N), H.mat(H._p(c), N), math=3,
                           accumulate=2, split_k=sp)
        ms = timeit(f, 5)
        print(f"TN M{M} N{N} K{K} split{sp}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TF", flush=True)
    print("auto split", H._split_for(M, N, K))
