"""Victims beside the matrix-core attention kernel (eager, second stream), 100 runs each."""
import os, sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H
dev = torch.device("cuda:0")
H.set_math("bf16x3")
g = torch.Generator().manual_seed(0)
B, L, Cin = 64, 1024, 64
x = torch.randn(B, L, Cin, generator=g).to(dev)
x3 = torch.randn(512, 256, 128, generator=g).to(dev)
M, N, K = 66048, 512, 128
a = torch.randn(M, K, device=dev); bw = torch.randn(N, K, device=dev)
qkv = torch.randn(512, 129, 384, device=dev); pad = torch.zeros(512, 129, dtype=torch.uint8, device=dev)
xl = torch.randn(66048, 128, device=dev); gam = torch.ones(128, device=dev); bet = torch.zeros(128, device=dev)


def gemm_nt():
    c = torch.empty(M, N, device=dev)
    H.gemm(0, M, N, K, H.mat(H._p(a), K), H.mat(H._p(bw), K), H.mat(H._p(c), N))
    return c


def ln():
    with torch.no_grad():
        return H.layer_norm(xl, gam, bet)


victims0 = {"fft_rows 1536": lambda: H.fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, (9, 1)),
           "fft_rows 2048": lambda: H.fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, 11),
           "fft_rows 384": lambda: H.fft_rows_fwd(x3, None, 0, 256 * 128, 128, 0, 512, 256, 128, 0, (7, 1)),
           "fft_rows 512": lambda: H.fft_rows_fwd(x3, None, 0, 256 * 128, 128, 0, 512, 256, 128, 0, 9),
           "gemm_x3 NT": gemm_nt, "layernorm": ln}
victims = victims0
side = torch.cuda.Stream()
for vn, vf in victims.items():
    ref = vf()
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for it in range(100):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.no_grad():
                for _ in range(4):
                    H.mha(qkv, pad, 8, 0.0, False)
        out = vf()
        torch.cuda.synchronize()
        e = float((out - ref).abs().max())
        bad += e != 0.0
        worst = max(worst, e)
    print(f"[{os.environ.get('TAG', 'default')}] victim {vn:14s} beside attention (mfma): {bad:3d} of 100 differ, worst {worst:.3e}", flush=True)
