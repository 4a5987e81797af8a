"""A few launches of gemm_x3_kernel at 512 workgroups (two per CU), K = 4096, for counter passes (tools/gpu_x3_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')
H.set_math("bf16x3")
M, N, K = 4096, 2048, 4096
c = torch.zeros(M, N, device=dev)
a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
for _ in range(3):
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c), N))
bt = torch.randn(K, N, device=dev)
for _ in range(3):
    H.gemm(H.AC_GEMM_NN, M, N, K, H.mat(H._p(a), K), H.mat(H._p(bt), N), H.mat(H._p(c), N))
at = torch.randn(K, M, device=dev)
for _ in range(3):
    H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(at), M), H.mat(H._p(bt), N), H.mat(H._p(c), N))
torch.cuda.synchronize()
