"""One 'same' Conv1d of the SpectraNet bank, forward + input gradient + weight gradient, through the frequency
domain (ac_fft.hip + ac_gemm_batched) and through the direct split-bf16 window kernels, on the shapes of the default
stages.  python tools/bench_fftconv.py"""
import sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H   # noqa: E402

dev = torch.device("cuda:0")
H.set_math("bf16x3")


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(512, 1024, 64, 128, 251), (512, 1024, 64, 128, 31), (512, 256, 128, 256, 61), (512, 256, 128, 256, 15),
          (512, 64, 256, 512, 31), (512, 64, 256, 512, 11), (512, 16, 512, 1024, 13), (512, 16, 512, 1024, 7)]
for (B, L, Cin, Cout, k) in shapes:
    x = torch.randn(B, L, Cin, device=dev, requires_grad=True)
    w = (torch.randn(Cout, k * Cin, device=dev) / (k * Cin) ** 0.5).requires_grad_()
    b = torch.zeros(Cout, device=dev, requires_grad=True)
    g = torch.randn(B, L, Cout, device=dev)
    res = {}
    for name, on in (("fft", True), ("direct", False)):
        H._FFTCONV, H._FFT_FORCE = on, on

        def step():
            y = H.conv_group1d(x, (k,), [w], [b])
            y.backward(g)
            x.grad = w.grad = b.grad = None
        res[name] = timed(step)
    H._FFTCONV, H._FFT_FORCE = True, False
    flop = 6.0 * B * L * Cin * Cout * k
    print(f"B={B} L={L} {Cin}->{Cout} k={k}: plan (logn, blocks, step) = {H.fft_plan(L, k)}  fft {res['fft']:.3f} ms  "
          f"direct {res['direct']:.3f} ms ({flop / res['direct'] / 1e9:.0f} TF)  rule says {'fft' if H.fftconv_covered(B, L, Cin, Cout, k) else 'direct'}",
          flush=True)
