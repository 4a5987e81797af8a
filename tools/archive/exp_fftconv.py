"""Probe for the next round, NOT product code: what an FFT formulation of SpectraNet's long-tap convolutions would cost
and how exact it is (spectranet.py:18-20; stage 2: Conv1d(64 -> 128, k = 251) on L = 1024, B = 512 = 2.16 TFLOP direct per
product, three products per step = 12 of the step's 44 ms on the split-bf16 window / weight-gradient kernels).

  forward      Y_f[b, co] = sum_ci X_f[b, ci] W_f[ci, co]          per frequency f (N/2 + 1 of them, N = 2048)
  input grad   dX_f[b, ci] = sum_co dY_f[b, co] conj(W_f[ci, co])
  weight grad  dW_f[ci, co] = sum_b conj(X_f[b, ci]) dY_f[b, co]
i.e. three batched complex products of 34 GFLOP each instead of 2 156 GFLOP, plus real FFTs of the operands.  This
script times that pipeline with torch.fft / torch.matmul on the GPU (vendor FFT and BLAS libraries: a floor estimate of
the BYTES such a path moves, not the product's kernels) and measures its error against torch's conv1d in fp64.
    python tools/exp_fftconv.py [B=512]"""
import sys
import time

import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (L, Cin, Cout, k) in ((1024, 64, 128, 251), (256, 128, 256, 61), (4096, 1, 64, 1021)):
    N = 1
    while N < L + k - 1:
        N *= 2
    x = torch.randn(B, Cin, L, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
    go = torch.randn(B, Cout, L, device=dev)
    p = k // 2

    def fwd_bwd():
        Xf = torch.fft.rfft(x, n=N)                                   # [B, Cin, F]
        Wf = torch.fft.rfft(w.flip(-1), n=N)                          # correlation = convolution with flipped taps
        Yf = torch.einsum("bif,oif->bof", Xf, Wf)
        y = torch.fft.irfft(Yf, n=N)[..., k - 1 - p:k - 1 - p + L]
        Gf = torch.fft.rfft(F.pad(go, (k - 1 - p, N - L - (k - 1 - p))), n=N)
        dXf = torch.einsum("bof,oif->bif", Gf, Wf.conj())
        dx = torch.fft.irfft(dXf, n=N)[..., :L]
        dWf = torch.einsum("bif,bof->oif", Xf.conj(), Gf)
        dw = torch.fft.irfft(dWf, n=N)[..., :k].flip(-1)
        return y, dx, dw

    y, dx, dw = fwd_bwd()
    x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
    y64 = F.conv1d(x64, w64, padding=p)
    y64.backward(go.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    errs = (rel(y, y64.detach()), rel(dx, x64.grad), rel(dw, w64.grad))
    for _ in range(2):
        fwd_bwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        fwd_bwd()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    direct = 3 * 2.0 * B * L * Cout * Cin * k
    print(f"L={L} Cin={Cin} Cout={Cout} k={k} N={N}: fwd+dx+dw {ms:.2f} ms through torch.fft + complex einsum "
          f"(direct: {direct / 1e12:.2f} TFLOP = {direct / 520e12 * 1e3:.2f} ms at 520 TFLOP/s); "
          f"max rel err vs fp64 conv1d: y {errs[0]:.1e} dx {errs[1]:.1e} dw {errs[2]:.1e}", flush=True)
