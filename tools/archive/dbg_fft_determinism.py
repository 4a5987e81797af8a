"""Run-to-run determinism of the frequency-domain conv products (no atomics anywhere on that path)."""
import sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H   # noqa: E402

dev = torch.device("cuda:0")
H.set_math("bf16x3")
for (B, L, Cin, Cout, k) in [(512, 1024, 64, 128, 251), (512, 256, 128, 256, 61), (512, 1024, 64, 128, 31), (512, 64, 256, 512, 31), (512, 16, 512, 1024, 13)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
    dy = torch.randn(B, L, Cout, generator=g).to(dev)
    outs = []
    for it in range(3):
        out = torch.zeros(B, L, Cout, device=dev)
        saved = H.fftconv_forward(x, w, B, L, Cin, Cout, k, out, Cout, 0, None)
        dx = torch.empty(B, L, Cin, device=dev)
        dw = torch.zeros(Cout, k * Cin, device=dev)
        H.fftconv_backward(saved, dy, None, 0, L * Cout, Cout, 0, B, L, Cin, Cout, k, dx, False, dw)
        torch.cuda.synchronize()
        outs.append((out.clone(), dx.clone(), dw.clone(), saved[1].clone()))
        junk = torch.randn(64 << 20, device=dev)   # dirty the allocator's blocks between runs
        del junk
    for i in (1, 2):
        print((B, L, Cin, Cout, k), H.fft_plan(L, k), "run", i, [bool(torch.equal(a, b)) for a, b in zip(outs[0], outs[i])],
              [float((a - b).abs().max()) for a, b in zip(outs[0], outs[i])])

    # against the direct window kernels
    xg = x.clone().requires_grad_(); wg = w.clone().requires_grad_(); bg = torch.zeros(Cout, device=dev, requires_grad=True)
    H._FFTCONV = False
    y = H.conv_group1d(xg, (k,), [wg], [bg]); y.backward(dy)
    H._FFTCONV = True
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print("   vs direct: y %.2e dx %.2e dw %.2e" % (rel(outs[0][0], y.detach()), rel(outs[0][1], xg.grad), rel(outs[0][2], wg.grad)), flush=True)
