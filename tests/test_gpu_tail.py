"""Fused tail of a pooled SpectraNetBlock (csrc/ac_tail.hip; spectranet.py:31-40: LayerNorm + GELU + 1x1 conv +
MaxPool(4)): the three kernels on their own against torch in fp64, and the whole block (conv bank + tail, forward and
every gradient) against the unfused kernels and against fp64."""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (rows, K, N): one block; a few blocks; more blocks than resident workgroups (every workgroup walks several)
SHAPES = [(32, 192, 64), (256, 192, 64), (32 * 2500, 192, 64), (96, 384, 128), (32 * 700, 384, 128)]


@pytest.fixture
def math_mode(request):
    from applecider_amd import hipops as H
    H.set_math(request.param)
    yield request.param
    H.set_math("f32")


def _l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def _inputs(rows, K, N):
    g = torch.Generator().manual_seed(rows + K)
    ycat = torch.randn(rows, K, generator=g) * 2 + 0.3
    gam, bet = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    dpool = torch.randn(rows // 4, N, generator=g)
    return ycat, gam, bet, w, b, dpool


def _reference(ycat, gam, bet, w, b, dpool):
    y64 = ycat.double().requires_grad_()
    g64, b64, w64, bb64 = (t.double().requires_grad_() for t in (gam, bet, w, b))
    z = F.gelu(F.layer_norm(y64, (y64.shape[1],), g64, b64, 1e-5))
    out = z @ w64.t() + bb64
    pooled, idx = F.max_pool1d(out.t().unsqueeze(0), 4, return_indices=True)
    pooled, idx = pooled[0].t(), idx[0].t() % 4
    return y64, g64, b64, w64, bb64, out, pooled, idx


def _split(t):
    from applecider_amd import hipops as H
    return H.split16(t.contiguous())


@pytest.mark.parametrize("rows,K,N", SHAPES)
def test_spectail_fwd_vs_fp64(dev, rows, K, N):
    from applecider_amd import _lib, hipops as H
    ycat, gam, bet, w, b, _ = _inputs(rows, K, N)
    _, _, _, _, _, out, pr, ir = _reference(ycat, gam, bet, w, b, None)
    d = lambda t: t.to(dev)
    yd, gd, bd, wd, bbd = d(ycat), d(gam), d(bet), d(w), d(b)
    assert _lib.load().ac_spectail_supported(rows, K, N) == 1
    assert _lib.load().ac_spectail_supported(rows + 1, K, N) == 0 and _lib.load().ac_spectail_supported(rows, 768, 256) == 0
    wh, wl = _split(wd)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    pooled = torch.full((rows // 4, N), float("nan"), device=dev)
    idx = torch.full((rows // 4, N), 255, device=dev, dtype=torch.uint8)
    _lib.check(_lib.load().ac_spectail_fwd(H._p(yd), H._p(gd), H._p(bd), 1e-5, H._p(wh), H._p(wl), H._p(bbd), H._p(mean),
                                           H._p(rstd), H._p(pooled), H._p(idx), rows, K, N, H._stream()), "ac_spectail_fwd")
    torch.cuda.synchronize()
    mu = ycat.double().mean(1)
    assert float((mean.cpu().double() - mu).abs().max()) <= 2e-6
    assert _l2(rstd, 1.0 / torch.sqrt(ycat.double().var(1, unbiased=False) + 1e-5)) <= 1e-6
    assert _rel(pooled, pr.detach()) <= 5e-5                                             # split-bf16 product
    srt = out.detach().reshape(rows // 4, 4, N).sort(1).values
    clear = (srt[:, 3] - srt[:, 2]) > 1e-3                                                # no near tie in the window
    assert torch.equal(idx.cpu().long()[clear], ir[clear])
    assert int(idx.max()) <= 3 and clear.float().mean() > 0.98


@pytest.mark.parametrize("seg", [False, True])
@pytest.mark.parametrize("rows,K,N", SHAPES)
def test_spectail_bwd_vs_fp64(dev, rows, K, N, seg):
    """Both backward kernels from the forward kernel's own statistics and arg-max: (hi + lo) planes of d ycat, d gamma,
    d beta, column sums, d w — against autograd in fp64 with the gradient routed through the SAME windows' positions."""
    from applecider_amd import _lib, hipops as H
    ycat, gam, bet, w, b, dpool = _inputs(rows, K, N)
    d = lambda t: t.to(dev)
    yd, gd, bd, wd, bbd, dpd = d(ycat), d(gam), d(bet), d(w), d(b), d(dpool)
    wh, wl = _split(wd)
    wth, wtl = _split(wd.t())
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    pooled = torch.empty(rows // 4, N, device=dev)
    idx = torch.empty(rows // 4, N, device=dev, dtype=torch.uint8)
    lib = _lib.load()
    _lib.check(lib.ac_spectail_fwd(H._p(yd), H._p(gd), H._p(bd), 1e-5, H._p(wh), H._p(wl), H._p(bbd), H._p(mean),
                                   H._p(rstd), H._p(pooled), H._p(idx), rows, K, N, H._stream()), "ac_spectail_fwd")
    # reference with the kernel's routing: out[4 i + idx] receives d pooled
    y64, g64, b64, w64, bb64, out, _, _ = _reference(ycat, gam, bet, w, b, dpool)
    dout = torch.zeros(rows // 4, 4, N, dtype=torch.float64)
    dout.scatter_(1, idx.cpu().long().unsqueeze(1), dpool.double().unsqueeze(1))
    out.backward(dout.reshape(rows, N))
    # zero-padded layout: "samples" of seg_len rows, pad rows in front and behind
    L = 32 if rows % 64 else 64
    pad = 5
    Lp = L + 2 * pad + 3
    if seg:
        planes = torch.zeros(2, rows // L, Lp, K, device=dev, dtype=torch.bfloat16)
        sl, sp, so = L, Lp, pad
    else:
        planes = torch.zeros(2, rows, K, device=dev, dtype=torch.bfloat16)
        sl = sp = so = 0
    dgam, dbet, dxs = torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    dw = torch.zeros(N, K, device=dev)
    for rep in range(2):   # accumulating outputs: two launches = twice the sums
        _lib.check(lib.ac_spectail_bwd_dx(H._p(yd), H._p(mean), H._p(rstd), H._p(gd), H._p(bd), H._p(dpd), H._p(idx),
                                          H._p(wth), H._p(wtl), H._p(planes[0]), H._p(planes[1]), sl, sp, so, H._p(dgam),
                                          H._p(dbet), H._p(dxs), rows, K, N, H._stream()), "ac_spectail_bwd_dx")
        _lib.check(lib.ac_spectail_bwd_dw(H._p(yd), H._p(mean), H._p(rstd), H._p(gd), H._p(bd), H._p(dpd), H._p(idx),
                                          H._p(dw), rows, K, N, H._stream()), "ac_spectail_bwd_dw")
    torch.cuda.synchronize()
    dx = planes[0].float() + planes[1].float()
    if seg:
        assert float(dx[:, :pad].abs().max()) == 0.0 and float(dx[:, pad + L:].abs().max()) == 0.0   # pads untouched
        dx = dx[:, pad:pad + L].reshape(rows, K)
    assert _rel(dx, y64.grad) <= 5e-5
    assert _rel(dgam, 2 * g64.grad) <= 5e-5 and _rel(dbet, 2 * b64.grad) <= 5e-5
    # (column sums cancel: error against the sum of magnitudes)
    assert float((dxs.cpu().double() - 2 * y64.grad.sum(0)).abs().max() / (2 * y64.grad.abs().sum(0).max())) <= 2e-5
    assert _rel(dw, 2 * w64.grad) <= 5e-5


@pytest.mark.parametrize("math_mode", ["bf16x3"], indirect=True)
@pytest.mark.parametrize("B,L,Cin,Cout,ks,needs_dx", [(2, 256, 64, 128, (3, 31, 251), True), (3, 128, 64, 128, (3, 31, 251), True),
                                                      (2, 2048, 1, 64, (3, 61, 1021), False), (2, 256, 64, 128, (3, 31, 251), False)])
def test_block_with_fused_tail_vs_unfused_and_fp64(dev, math_mode, B, L, Cin, Cout, ks, needs_dx):
    from applecider_amd import hipops as H
    gen = torch.Generator().manual_seed(7 + L)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64).requires_grad_(needs_dx)
    ws = [(torch.randn(Cout, Cin, k, generator=gen, dtype=torch.float64) / math.sqrt(Cin * k)).requires_grad_() for k in ks]
    bs = [torch.randn(Cout, generator=gen, dtype=torch.float64).requires_grad_() for _ in ks]
    Ncat = 3 * Cout
    gam = (1 + 0.1 * torch.randn(Ncat, generator=gen, dtype=torch.float64)).requires_grad_()
    bet = (0.1 * torch.randn(Ncat, generator=gen, dtype=torch.float64)).requires_grad_()
    pw = (torch.randn(Cout, Ncat, generator=gen, dtype=torch.float64) / math.sqrt(Ncat)).requires_grad_()
    pb = torch.randn(Cout, generator=gen, dtype=torch.float64).requires_grad_()
    y = torch.cat([F.conv1d(x, w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)
    zz = F.gelu(F.layer_norm(y.permute(0, 2, 1), (Ncat,), gam, bet, 1e-5))
    out = F.max_pool1d((zz @ pw.t() + pb).permute(0, 2, 1), 4).permute(0, 2, 1)
    go = torch.randn(*out.shape, generator=gen, dtype=torch.float64)
    out.backward(go)
    res = {}
    assert H._FUSED_TAIL and H.tail_covered(B, L, Cin, Cout)
    for fused in (True, False):
        H._FUSED_TAIL = fused
        assert H.tail_covered(B, L, Cin, Cout) == fused
        try:
            xd = x.detach().float().permute(0, 2, 1).contiguous().to(dev).requires_grad_(needs_dx)
            wd = [w.detach().float().permute(0, 2, 1).reshape(Cout, -1).contiguous().to(dev).requires_grad_() for w in ws]
            bd = [b.detach().float().to(dev).requires_grad_() for b in bs]
            gd, btd = gam.detach().float().to(dev).requires_grad_(), bet.detach().float().to(dev).requires_grad_()
            pwd, pbd = pw.detach().float().to(dev).requires_grad_(), pb.detach().float().to(dev).requires_grad_()
            if fused:
                od = H.conv_group1d(xd, ks, wd, bd, ln=(gd, btd, 1e-5), tail=(pwd, pbd))
            else:
                od = H.maxpool4(H.linear(H.conv_group1d(xd, ks, wd, bd, ln=(gd, btd, 1e-5)), pwd, pbd))
            od.backward(go.float().to(dev))
            torch.cuda.synchronize()
        finally:
            H._FUSED_TAIL = True
        res[fused] = [od.detach()] + ([xd.grad] if needs_dx else []) + [w.grad for w in wd] + [b.grad for b in bd] + \
                     [gd.grad, btd.grad, pwd.grad, pbd.grad]
    want = [out.detach()] + ([x.grad.permute(0, 2, 1)] if needs_dx else []) + \
           [w.grad.permute(0, 2, 1).reshape(Cout, -1) for w in ws] + [b.grad for b in bs] + [gam.grad, bet.grad, pw.grad, pb.grad]
    for i, (a, u, w_) in enumerate(zip(res[True], res[False], want)):
        assert a.shape == u.shape
        assert _l2(a, u) <= 5e-5, ("vs unfused", i, _l2(a, u))
        assert _l2(a, w_) <= 2e-4, ("vs fp64", i, _l2(a, w_))


def test_tail_is_on_the_default_path(dev):
    """The fused tail is what a training step of the default SpectraNet runs for its first two stages in the benchmarked
    mode (no environment switch), and it is not taken in the exact-fp32 mode."""
    from applecider_amd import hipops as H
    H.set_math("bf16x3")
    try:
        assert H.tail_covered(512, 4096, 1, 64) and H.tail_covered(512, 1024, 64, 128)
        assert not H.tail_covered(512, 256, 128, 256) and not H.tail_covered(512, 64, 256, 512)
        assert H.tail_covered(2, 4096, 1, 64) and H.tail_covered(2, 1024, 64, 128)
    finally:
        H.set_math("f32")
    assert not H.tail_covered(512, 4096, 1, 64)
