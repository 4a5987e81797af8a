"""Times the frequency-domain conv products (hipops.fftconv_forward / fftconv_backward) piece by piece against the
direct split-bf16 window kernels on the SpectraNet shapes.  python tools/bench_fftconv.py [f32|bf16x3]"""
import sys
import torch
sys.path.insert(0, ".")
from applecider_amd import _lib, hipops as H   # noqa: E402

dev = torch.device("cuda:0")
H.set_math("bf16x3")
if len(sys.argv) > 1:
    H._FFT_MATH = _lib.MATH_F32 if sys.argv[1] == "f32" else _lib.MATH_BF16X3


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


for (B, L, Cin, Cout, k) in [(512, 1024, 64, 128, 251), (512, 1024, 64, 128, 31), (512, 256, 128, 256, 61),
                             (512, 256, 128, 256, 15), (512, 64, 256, 512, 31), (512, 64, 256, 512, 11)]:
    logn = H.fft_logn(L, k)
    F = (1 << (logn - 1)) + 1
    x = torch.randn(B, L, Cin, device=dev)
    w = torch.randn(Cout, k * Cin, device=dev) / (k * Cin) ** 0.5
    Ncat = 3 * Cout
    out = torch.empty(B, L, Ncat, device=dev)
    dyc = torch.randn(B, L, Ncat, device=dev)
    dx = torch.empty(B, L, Cin, device=dev)
    dw = torch.zeros(Cout, k * Cin, device=dev)
    col = 2 * Cout
    saved = H.fftconv_forward(x, w, B, L, Cin, Cout, k, out, Ncat, col, None)
    xf, hb, _ = saved
    yf = torch.empty(F, B, 2 * Cout, device=dev)
    gf = torch.empty(F, B, 2 * Cout, device=dev)
    dxf = torch.empty(F, B, 2 * Cin, device=dev)
    mp = torch.empty(F, 2 * Cout, 2 * Cin, device=dev)
    lib, tw, s = _lib.load(), H._fft_tw(logn, dev), H._stream()
    t = {}
    t["rows_fwd x"] = timed(lambda: H.fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, logn))
    t["taps_fwd"] = timed(lambda: lib.ac_fft_taps_fwd(H._p(w), Cout, Cin, k, logn, H._p(tw), H._p(hb), s))
    t["prod y"] = timed(lambda: H.gemm_batched(H.AC_GEMM_NT, B, 2 * Cout, 2 * Cin, H.mat(H._p(xf), 2 * Cin), H.mat(H._p(hb), 2 * Cin),
                                               H.mat(H._p(yf), 2 * Cout), F, B * 2 * Cin, 4 * Cout * Cin, B * 2 * Cout))
    t["rows_inv y"] = timed(lambda: H.fft_rows_inv(yf, B, Cout, logn, out, L * Ncat, Ncat, col, L, k - 1 - k // 2, None, False))
    t["rows_fwd dy"] = timed(lambda: H.fft_rows_fwd(dyc, None, 0, L * Ncat, Ncat, col, B, L, Cout, k - 1 - k // 2, logn))
    t["prod dx"] = timed(lambda: H.gemm_batched(H.AC_GEMM_NN, B, 2 * Cin, 2 * Cout, H.mat(H._p(gf), 2 * Cout), H.mat(H._p(hb), 2 * Cin),
                                                H.mat(H._p(dxf), 2 * Cin), F, B * 2 * Cout, 4 * Cout * Cin, B * 2 * Cin))
    t["rows_inv dx"] = timed(lambda: H.fft_rows_inv(dxf, B, Cin, logn, dx, L * Cin, Cin, 0, L, 0, None, True))
    t["prod dw"] = timed(lambda: H.gemm_batched(H.AC_GEMM_TN, 2 * Cout, 2 * Cin, B, H.mat(H._p(gf), 2 * Cout), H.mat(H._p(xf), 2 * Cin),
                                                H.mat(H._p(mp), 2 * Cin), F, B * 2 * Cout, B * 2 * Cin, 4 * Cout * Cin))
    t["taps_inv"] = timed(lambda: lib.ac_fft_taps_inv(H._p(mp), Cout, Cin, k, logn, H._p(tw), H._p(dw), s))
    fwd = timed(lambda: H.fftconv_forward(x, w, B, L, Cin, Cout, k, out, Ncat, col, None))
    bwd = timed(lambda: H.fftconv_backward(saved, dyc, None, 0, L * Ncat, Ncat, col, B, L, Cin, Cout, k, dx, True, dw))
    direct = 3 * 2.0 * B * L * Cout * Cin * k
    gb = lambda n: n * 4 / 1e9
    print(f"B={B} L={L} {Cin}->{Cout} k={k} N={1 << logn}: forward {fwd:.3f} ms, backward {bwd:.3f} ms, trio {fwd + bwd:.3f} ms "
          f"(direct {direct / 1e12:.2f} TFLOP = {direct / 520e9:.2f} ms at 520 TF)", flush=True)
    sizes = {"rows_fwd x": gb(x.numel() + xf.numel()), "taps_fwd": gb(w.numel() + hb.numel()),
             "prod y": gb(xf.numel() + hb.numel() + yf.numel()), "rows_inv y": gb(yf.numel() + B * L * Cout),
             "rows_fwd dy": gb(B * L * Cout + gf.numel()), "prod dx": gb(gf.numel() + hb.numel() + dxf.numel()),
             "rows_inv dx": gb(dxf.numel() + 2 * dx.numel()), "prod dw": gb(gf.numel() + xf.numel() + mp.numel()),
             "taps_inv": gb(mp.numel() + 2 * dw.numel())}
    for name, ms in t.items():
        print(f"    {name:12s} {ms:7.3f} ms  {sizes[name]:6.3f} GB  {sizes[name] / ms:6.2f} TB/s", flush=True)
