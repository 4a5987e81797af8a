import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')
M = N = 2048; K = 8192
c = torch.zeros(M, N, device=dev)
a, b = H.cast16(torch.randn(M, K, device=dev)), H.cast16(torch.randn(N, K, device=dev))
for _ in range(3):
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c), N), math=2)
at, bt = H.cast16(torch.randn(K, M, device=dev)), H.cast16(torch.randn(K, N, device=dev))
for _ in range(3):
    H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(at), M), H.mat(H._p(bt), N), H.mat(H._p(c), N), math=2)
torch.cuda.synchronize()
