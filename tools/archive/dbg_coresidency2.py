"""Who is the victim, who the culprit?  Kernel family X on the main stream (checked against its solo result) while family Y
runs on a second stream."""
import os, sys
import torch
sys.path.insert(0, ".")
from applecider_amd import _lib, hipops as H

dev = torch.device("cuda:0")
H.set_math("bf16x3")
g = torch.Generator().manual_seed(0)
B, L, Cin, Cout, k = 64, 1024, 64, 128, 251
x = torch.randn(B, L, Cin, generator=g).to(dev)
w = (torch.randn(Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
M, N, K = 66048, 512, 128
a = torch.randn(M, K, device=dev); bw = torch.randn(N, K, device=dev)
qkv = torch.randn(512, 129, 384, device=dev); pad = torch.zeros(512, 129, dtype=torch.uint8, device=dev)
xl = torch.randn(66048, 128, device=dev); gam = torch.ones(128, device=dev); bet = torch.zeros(128, device=dev)


def fft_rows(size):
    return lambda: (H.fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, size),)


def fft_taps(size):
    return lambda: (H.fft_taps_fwd(w, Cout, Cin, k, size),)


def gemm_nt():
    c = torch.empty(M, N, device=dev)
    H.gemm(0, M, N, K, H.mat(H._p(a), K), H.mat(H._p(bw), K), H.mat(H._p(c), N))
    return (c,)


def attn(mfma=True):
    def f():
        H._MHA_MFMA = mfma
        with torch.no_grad():
            o = H.mha(qkv, pad, 8, 0.0, False)
        H._MHA_MFMA = True
        return (o,)
    return f


def ln():
    with torch.no_grad():
        return (H.layer_norm(xl, gam, bet),)


victims = {"fft_rows 1536": fft_rows((9, 1)), "fft_rows 2048": fft_rows(11), "fft_rows 512x": fft_rows(9), "fft_taps 1536": fft_taps((9, 1)),
           "gemm_x3 NT": gemm_nt, "attention": attn(), "layernorm": ln}
culprits = {"attention (mfma)": attn(True), "attention (scalar fp32)": attn(False), "fft_rows 1536": fft_rows((9, 1)), "gemm_x3 NT": gemm_nt}
side = torch.cuda.Stream()
for vn, vf in victims.items():
    ref = vf()
    torch.cuda.synchronize()
    for cn, cf in culprits.items():
        if cn.split()[0] == vn.split()[0] and not vn.startswith("fft"):
            continue
        bad, worst = 0, 0.0
        for it in range(20):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(6):
                    cf()
            out = vf()
            torch.cuda.synchronize()
            e = max(float((o - r).abs().max()) for o, r in zip(out, ref))
            bad += e != 0.0
            worst = max(worst, e)
        print(f"victim {vn:14s} beside {cn:24s}: {bad:2d} of 20 differ, worst {worst:.3e}", flush=True)
