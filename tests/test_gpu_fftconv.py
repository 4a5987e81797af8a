"""Frequency-domain form of the long-tap Conv1d products (csrc/ac_fft.hip + ac_gemm_batched) against torch's conv1d
(spectranet.py:18-20: 'same' Conv1d, padding k//2) and its backward in fp64: transforms alone (vs torch.fft), then
forward, input gradient and weight gradient through the C ABI."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("logn", [5, 6, 7, 8, 9, 10, 11, (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1),
                                  (3, 2), (4, 2), (5, 2), (6, 2), (7, 2)])
@pytest.mark.parametrize("planes", [False, True])
def test_fft_rows_roundtrip_and_spectrum(dev, logn, planes):
    """rows -> spectrum equals torch.fft.rfft of the shifted, zero-padded rows; spectrum -> rows returns them.
    Sizes: N = 2^logn (32 ... 2048) and, as (logm, a), N = 3^a * 2^logm (24 ... 1536, 72 ... 1152)."""
    from applecider_amd import hipops as H
    N = H._fft_size(logn)[2]
    B, Cn, Ctot, col = 3, 32, 80, 16
    L, shift = N - 11, 5
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, L, Ctot, generator=g)
    xd = x.to(dev)
    if planes:
        hi, lo = H.split16(xd)
        spec = H.fft_rows_fwd(hi, lo, 0, L * Ctot, Ctot, col, B, L, Cn, shift, logn)
        x = (hi.float() + lo.float()).cpu()
    else:
        spec = H.fft_rows_fwd(xd, None, 0, L * Ctot, Ctot, col, B, L, Cn, shift, logn)
    seq = torch.zeros(B, N, Cn, dtype=torch.float64)
    seq[:, shift:shift + L] = x[:, :, col:col + Cn].double()
    want = torch.fft.rfft(seq, dim=1)                                   # [B, F, Cn]
    got = spec.cpu().double().reshape(N // 2 + 1, B, Cn, 2)
    got = torch.complex(got[..., 0], got[..., 1]).permute(1, 0, 2)
    assert float((got - want).abs().max() / want.abs().max()) <= 2e-6
    # inverse: back into a wider tensor at another column, with bias, then accumulated once more
    out = torch.full((B, L, Ctot), 7.0, device=dev)
    bias = torch.randn(Cn, generator=g).to(dev)
    H.fft_rows_inv(spec, B, Cn, logn, out, L * Ctot, Ctot, col + 16, L, shift, bias, False)
    ref = x[:, :, col:col + Cn].double() + bias.cpu().double()
    assert _rel(out[:, :, col + 16:col + 16 + Cn], ref) <= 2e-6
    assert float((out[:, :, :col + 16] - 7.0).abs().max()) == 0.0       # neighbours untouched
    H.fft_rows_inv(spec, B, Cn, logn, out, L * Ctot, Ctot, col + 16, L, shift, None, True)
    assert _rel(out[:, :, col + 16:col + 16 + Cn], ref + x[:, :, col:col + Cn].double()) <= 2e-6


@pytest.mark.parametrize("logn,B,Cn", [((7, 2), 70, 128),    # 1152 points, 8 sequences per workgroup: 560 tiles
                                       (9, 150, 128),         # 512 points, 32 sequences: 300 tiles (256 + a ragged round)
                                       ((5, 1), 330, 256),    # 96 points, 128 sequences: 330 tiles
                                       ((5, 2), 101, 48)])    # 288 points, 8 sequences x 512 threads: 303 tiles, rows % 8 != 0
@pytest.mark.parametrize("planes", [False, True])
def test_fft_rows_many_tiles_per_workgroup(dev, logn, B, Cn, planes):
    """More tiles than workgroups (ac_fft.hip rows_grid): a workgroup walks tiles t, t + grid, ... with the loads of the
    next one issued before the passes of the current one.  Spectrum against torch.fft.rfft, rows back exactly."""
    from applecider_amd import hipops as H
    N = H._fft_size(logn)[2]
    L, shift = N - 7, 3
    g = torch.Generator().manual_seed(N + B)
    x = torch.randn(B, L, Cn, generator=g)
    xd = x.to(dev)
    if planes:
        hi, lo = H.split16(xd)
        spec = H.fft_rows_fwd(hi, lo, 0, L * Cn, Cn, 0, B, L, Cn, shift, logn)
        x = (hi.float() + lo.float()).cpu()
    else:
        spec = H.fft_rows_fwd(xd, None, 0, L * Cn, Cn, 0, B, L, Cn, shift, logn)
    seq = torch.zeros(B, N, Cn, dtype=torch.float64)
    seq[:, shift:shift + L] = x.double()
    want = torch.fft.rfft(seq, dim=1)
    got = spec.cpu().double().reshape(N // 2 + 1, B, Cn, 2)
    got = torch.complex(got[..., 0], got[..., 1]).permute(1, 0, 2)
    assert float((got - want).abs().max() / want.abs().max()) <= 2e-6
    out = torch.empty(B, L, Cn, device=dev)
    H.fft_rows_inv(spec, B, Cn, logn, out, L * Cn, Cn, 0, L, shift, None, False)
    assert _rel(out, x.double()) <= 2e-6


def _ref_conv(x, w, b, dy, k):
    """fp64 reference: y, dx, dw (tap-major [Cout, k*Cin]) of the 'same' Conv1d on channels-last x."""
    xt = x.double().permute(0, 2, 1).requires_grad_(True)
    Cout, Cin = w.shape[0], x.shape[2]
    wt = w.double().reshape(Cout, k, Cin).permute(0, 2, 1).contiguous().requires_grad_(True)
    y = F.conv1d(xt, wt, b.double(), padding=k // 2)
    gx, gw = torch.autograd.grad(y, (xt, wt), dy.double().permute(0, 2, 1))
    return y.permute(0, 2, 1), gx.permute(0, 2, 1), gw.permute(0, 2, 1).reshape(Cout, k * Cin)


@pytest.mark.parametrize("B,L,Cin,Cout,k", [(4, 1024, 64, 128, 251),     # SpectraNet stage 2: one sequence of 2048
                                            (2, 1024, 64, 128, 129),     # 3 windows of 512
                                            (2, 1024, 64, 128, 31),      # stage 2: 5 windows of 256
                                            (4, 256, 128, 256, 61),      # stage 3: one sequence of 512
                                            (4, 256, 128, 256, 15),      # stage 3: 6 windows of 64, 64-channel groups
                                            (8, 64, 256, 512, 31),       # stage 4: one sequence of 128
                                            (8, 64, 256, 512, 11),       # stage 4: 3 windows of 32
                                            (8, 16, 512, 1024, 13),      # stage 5: one sequence of 32
                                            (2, 900, 16, 32, 601),       # one sequence of 2048 (the 148 KB kernels)
                                            (3, 128, 16, 48, 99),        # odd batch, N = 256, 16-channel groups
                                            (3, 100, 16, 32, 21),        # 3 windows of 64, the last one partial
                                            (2, 40, 32, 16, 49)])        # L not a power of two, N = 64
@pytest.mark.parametrize("math", ["f32", "bf16x3"])
@pytest.mark.parametrize("radix3", [True, False, 9])
def test_fftconv_products(dev, B, L, Cin, Cout, k, math, radix3):
    """radix3 = False keeps the plans on power-of-two transform lengths (2048 points for stage 2's k = 251); 9 also
    admits the lengths 9 * 2^m (1152 points for it; the default), True stops at 3 * 2^m (1536)."""
    from applecider_amd import _lib, hipops as H
    H._FFT_RADIX3, H._FFT_RADIX9 = bool(radix3), radix3 == 9
    try:
        _products(dev, B, L, Cin, Cout, k, math)
    finally:
        H._FFT_RADIX3, H._FFT_RADIX9 = True, True


def _products(dev, B, L, Cin, Cout, k, math):
    from applecider_amd import _lib, hipops as H
    assert H.fft_plan(L, k) is not None
    g = torch.Generator().manual_seed(k + L)
    x = torch.randn(B, L, Cin, generator=g)
    w = torch.randn(Cout, k * Cin, generator=g) / (k * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    Ncat, col = Cout + 32, 16
    dyc = torch.randn(B, L, Ncat, generator=g)
    dy = dyc[:, :, col:col + Cout]
    y_ref, dx_ref, dw_ref = _ref_conv(x, w, b, dy, k)
    old = H._FFT_MATH
    H._FFT_MATH = _lib.MATH_F32 if math == "f32" else _lib.MATH_BF16X3
    try:
        xd, wd = x.to(dev), w.to(dev)
        out = torch.zeros(B, L, Ncat, device=dev)
        saved = H.fftconv_forward(xd, wd, B, L, Cin, Cout, k, out, Ncat, col, b.to(dev))
        tol = 5e-6 if math == "f32" else 5e-5
        assert _rel(out[:, :, col:col + Cout], y_ref) <= tol
        assert float(out[:, :, :col].abs().max()) == 0.0 and float(out[:, :, col + Cout:].abs().max()) == 0.0
        # gradients from fp32 rows of the concatenated gradient ...
        dyd = dyc.to(dev)
        dx = torch.empty(B, L, Cin, device=dev)
        dw = torch.zeros(Cout, k * Cin, device=dev)
        H.fftconv_backward(saved, dyd, None, 0, L * Ncat, Ncat, col, B, L, Cin, Cout, k, dx, False, dw)
        assert _rel(dx, dx_ref) <= tol
        assert _rel(dw, dw_ref) <= tol
        # ... and from zero-padded (hi, lo) planes as LayerNorm's backward writes them; dx / dw accumulate
        P = 5
        hi, lo = H._pad_rows_split(dyd, B, L, Ncat, P, L + 2 * P)
        H.fftconv_backward(saved, hi, lo, P * Ncat, (L + 2 * P) * Ncat, Ncat, col, B, L, Cin, Cout, k, dx, True, dw)
        assert _rel(dx, 2 * dx_ref) <= 2 * tol + 2e-5
        assert _rel(dw, 2 * dw_ref) <= 2 * tol + 2e-5
    finally:
        H._FFT_MATH = old


def test_fft_plan_and_refusals(dev):
    from applecider_amd import _lib, hipops as H
    # plans of the default SpectraNet stages (default_config.toml:104-114): (logn, blocks, rows a block advances)
    assert H.fft_plan(1024, 251) == (7, 2, 1, 1024) and H.fft_plan(256, 61) == (5, 2, 1, 256)      # 1152 = 9 * 128, 288 points
    assert H.fft_plan(64, 31) == (5, 1, 1, 64) and H.fft_plan(16, 13) == (3, 1, 1, 16)            # 96, 24 points
    H._FFT_RADIX9 = False
    try:
        assert H.fft_plan(1024, 251) == (9, 1, 1, 1024) and H.fft_plan(1024, 31) == (7, 1, 3, 354)  # 1536; 3 windows of 384
        assert H.fft_plan(256, 61) == (7, 1, 1, 256)
    finally:
        H._FFT_RADIX9 = True
    assert H.fft_plan(4096, 1021) is None
    H._FFT_RADIX3 = False
    try:
        assert H.fft_plan(1024, 251) == (11, 0, 1, 1024) and H.fft_plan(1024, 31) == (8, 0, 5, 226)
        assert H.fft_plan(256, 61) == (9, 0, 1, 256) and H.fft_logn(900, 601) == 11
    finally:
        H._FFT_RADIX3 = True
    lib = _lib.load()
    z = torch.zeros(8192, device=dev)
    s = H._stream()

    def desc(**kw):
        base = dict(rows=z, rows_lo=None, elem_off=0, spec=z, bias=None, batch_stride=64, row_stride=16, col_off=0, B=1, L=4,
                    Cn=16, size=6, blocks=1, step=0, shift=0, n_lo=0, n_hi=0, accumulate=0)
        base.update(kw)
        return H._fft_rows_desc(**base)
    import ctypes as C
    assert lib.ac_fft_rows_fwd(C.byref(desc()), s) == 0
    assert lib.ac_fft_rows_fwd(C.byref(desc(Cn=8)), s) == _lib.AC_EINVAL                    # C % 16
    assert lib.ac_fft_rows_fwd(C.byref(desc(L=60, shift=8)), s) == _lib.AC_EINVAL           # L + shift > N, one sequence
    assert lib.ac_fft_rows_fwd(C.byref(desc(L=4, blocks=2, step=1)), s) == _lib.AC_EINVAL   # blocks do not cover L
    assert lib.ac_fft_rows_inv(C.byref(desc(L=4, blocks=2, step=62, shift=8)), s) == _lib.AC_EINVAL   # step + shift > N
    assert lib.ac_fft_rows_fwd(C.byref(desc(size=(10, 1))), s) == _lib.AC_EINVAL            # 3 * 1024 > 2048 points
    assert lib.ac_fft_rows_fwd(C.byref(desc(size=(8, 2))), s) == _lib.AC_EINVAL             # 9 * 256 > 2048 points
    tw = H._fft_tw(6, dev)
    assert lib.ac_fft_taps_fwd(H._p(z), 2, 16, 65, 6, 0, H._p(tw), H._p(z), s) == _lib.AC_EINVAL    # k > N
    assert lib.ac_fft_taps_fwd(H._p(z), 2, 16, 5, 12, 0, H._p(tw), H._p(z), s) == _lib.AC_EINVAL    # logn > 11
