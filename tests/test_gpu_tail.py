"""Fused tail of a pooled SpectraNetBlock (csrc/ac_tail.hip: LayerNorm + GELU + 1x1 conv + MaxPool(4) in one forward
kernel, spectranet.py:31-40) against torch in fp64, and the whole block (conv bank + tail, forward and every gradient)
against the unfused kernels and against fp64."""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def math_mode(request):
    from applecider_amd import hipops as H
    H.set_math(request.param)
    yield request.param
    H.set_math("f32")


def _l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("R,K,N", [(256, 192, 64), (384, 384, 128), (128, 768, 256), (128, 1536, 512), (256, 64, 32)])
def test_ln_gelu_pw_pool_kernel_vs_fp64(dev, R, K, N):
    from applecider_amd import _lib, hipops as H
    g = torch.Generator().manual_seed(R + K)
    ycat = torch.randn(R, K, generator=g) * 2 + 0.3
    gam, bet = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    zr = F.gelu(F.layer_norm(ycat.double(), (K,), gam.double(), bet.double(), 1e-5))
    out = zr @ w.double().t() + b.double()
    pr, ir = F.max_pool1d(out.t().unsqueeze(0), 4, return_indices=True)
    pr, ir = pr[0].t(), ir[0].t() % 4                                   # [R / 4, N]
    d = lambda t: t.to(dev)
    z = torch.empty(R, K, device=dev)
    mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
    pooled, idx = torch.empty(R // 4, N, device=dev), torch.empty(R // 4, N, device=dev, dtype=torch.uint8)
    yd, gd, bd, wd, bbd = d(ycat), d(gam), d(bet), d(w), d(b)
    _lib.check(_lib.load().ac_ln_gelu_pw_pool_fwd(H._p(yd), K, H._p(gd), H._p(bd), 1e-5, H._p(wd), H._p(bbd), H._p(z),
                                                  H._p(mean), H._p(rstd), H._p(pooled), H._p(idx), R, K, N, H._stream()),
               "ac_ln_gelu_pw_pool_fwd")
    assert float((z.cpu().double() - zr).abs().max()) <= 5e-6
    mu = ycat.double().mean(1)
    assert float((mean.cpu().double() - mu).abs().max()) <= 1e-6
    assert _l2(rstd, 1.0 / torch.sqrt(ycat.double().var(1, unbiased=False) + 1e-5)) <= 1e-6
    assert float((pooled.cpu().double() - pr).abs().max() / pr.abs().max()) <= 5e-5          # split-bf16 product
    srt = out.reshape(R // 4, 4, N).sort(1).values
    clear = (srt[:, 3] - srt[:, 2]) > 1e-3                                                   # no near tie in the group
    assert torch.equal(idx.cpu().long()[clear], ir[clear])
    assert clear.float().mean() > 0.98


@pytest.mark.parametrize("math_mode", ["bf16x3"], indirect=True)
@pytest.mark.parametrize("B,L,Cin,Cout,ks", [(2, 256, 64, 128, (3, 31, 251)), (4, 64, 128, 64, (3, 11, 31)), (2, 512, 1, 64, (3, 61, 1021))])
def test_block_with_fused_tail_vs_unfused_and_fp64(dev, math_mode, B, L, Cin, Cout, ks):
    from applecider_amd import hipops as H
    gen = torch.Generator().manual_seed(7 + L)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64).requires_grad_(Cin != 1)
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
    for fused in (True, False):
        H._FUSED_TAIL = fused
        assert H.tail_covered(B, L, Ncat, Cout) == fused
        try:
            xd = x.detach().float().permute(0, 2, 1).contiguous().to(dev).requires_grad_(Cin != 1)
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
            H._FUSED_TAIL = False
        res[fused] = [od.detach()] + ([xd.grad] if Cin != 1 else []) + [w.grad for w in wd] + [b.grad for b in bd] + \
                     [gd.grad, btd.grad, pwd.grad, pbd.grad]
    want = [out.detach()] + ([x.grad.permute(0, 2, 1)] if Cin != 1 else []) + \
           [w.grad.permute(0, 2, 1).reshape(Cout, -1) for w in ws] + [b.grad for b in bs] + [gam.grad, bet.grad, pw.grad, pb.grad]
    for i, (a, u, w_) in enumerate(zip(res[True], res[False], want)):
        assert a.shape == u.shape
        assert _l2(a, u) <= 2e-5, ("vs unfused", i, _l2(a, u))
        assert _l2(a, w_) <= 2e-4, ("vs fp64", i, _l2(a, w_))
