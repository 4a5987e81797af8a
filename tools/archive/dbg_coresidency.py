"""Do the transform kernels stay exact while ANOTHER kernel family runs on a second stream (eager, no graph)?"""
import sys
import torch
sys.path.insert(0, ".")
from applecider_amd import _lib, hipops as H

dev = torch.device("cuda:0")
H.set_math("bf16x3")
g = torch.Generator().manual_seed(0)
B, L, Cin, Cout, k = 64, 1024, 64, 128, 251
x = torch.randn(B, L, Cin, generator=g).to(dev)
w = (torch.randn(Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
size = (9, 1)   # 1536 points


def mine():
    xf = H.fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, size)
    hb = H.fft_taps_fwd(w, Cout, Cin, k, size)
    return xf, hb


ref = mine()
torch.cuda.synchronize()
side = torch.cuda.Stream()
# foreign work
M, N, K = 66048, 512, 128
a = torch.randn(M, K, device=dev); bw = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
xl = torch.randn(66048, 128, device=dev); gam = torch.ones(128, device=dev); bet = torch.zeros(128, device=dev)
qkv = torch.randn(512, 129, 384, device=dev); pad = torch.zeros(512, 129, dtype=torch.uint8, device=dev)
img = torch.randn(512, 15, 15, 96, device=dev); dww = torch.randn(49, 96, device=dev); dwb = torch.zeros(96, device=dev)
foreign = {
    "nothing": lambda: None,
    "gemm_x3 NT": lambda: H.gemm(0, M, N, K, H.mat(H._p(a), K), H.mat(H._p(bw), K), H.mat(H._p(c), N)),
    "gemm_x3 NN": lambda: H.gemm(1, M, K, N, H.mat(H._p(c), N), H.mat(H._p(bw), K), H.mat(H._p(a), K)),
    "layernorm": lambda: H.layer_norm(xl, gam, bet),
    "attention": lambda: H.mha(qkv, pad, 8, 0.0, False),
    "dwconv7x7": lambda: H.dwconv7x7(img, dww, dwb),
}
for name, f in foreign.items():
    bad = 0
    worst = 0.0
    for it in range(30):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.no_grad():
                for _ in range(6):
                    f()
        out = mine()
        torch.cuda.synchronize()
        e = max(float((o - r).abs().max()) for o, r in zip(out, ref))
        bad += e != 0.0
        worst = max(worst, e)
    print(f"beside {name:12s}: {bad:2d} of 30 runs differ, worst abs diff {worst:.3e}", flush=True)
