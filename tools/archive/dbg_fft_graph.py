"""hipGraph replay of the frequency-domain conv products against the eager launches."""
import sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H   # noqa: E402

dev = torch.device("cuda:0")
H.set_math("bf16x3")
for (B, L, Cin, Cout, k) in [(8, 1024, 64, 128, 251), (8, 256, 128, 256, 61), (8, 64, 256, 512, 31)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
    dy = torch.randn(B, L, Cout, generator=g).to(dev)
    out = torch.zeros(B, L, Cout, device=dev)
    dx = torch.empty(B, L, Cin, device=dev)
    dw = torch.zeros(Cout, k * Cin, device=dev)

    def run():
        dw.zero_()
        saved = H.fftconv_forward(x, w, B, L, Cin, Cout, k, out, Cout, 0, None)
        H.fftconv_backward(saved, dy, None, 0, L * Cout, Cout, 0, B, L, Cin, Cout, k, dx, False, dw)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run(); run()
    torch.cuda.synchronize()
    ref = (out.clone(), dx.clone(), dw.clone())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    for it in range(3):
        out.zero_(); dx.zero_()
        gr.replay()
        torch.cuda.synchronize()
        print((B, L, Cin, Cout, k), "replay", it, [float((a - b).abs().max()) for a, b in zip(ref, (out, dx, dw))], flush=True)
