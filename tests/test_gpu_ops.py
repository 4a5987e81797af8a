"""Kernel-level parity: every HIP op (through the C ABI) against the PyTorch CPU fp32 op the
reference dispatches to.  Tolerances: fp32 MFMA path 2e-4 relative to the tensor's max (sum
order differs), bf16-input path 2e-2."""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, tol=2e-4, name=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, f"{name}: rel-to-max err {err:.3e} > {tol}"


def g(dev, *shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=gen) * scale)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 100), (1000, 96, 384), (64, 520, 1028),
                                   (257, 5, 64)])
@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
def test_gemm_modes(dev, M, N, K, mode):
    from applecider_amd import hipops as H
    a = g(dev, M, K, seed=1)
    b = g(dev, N, K, seed=2)
    ref = a @ b.t()
    ad, bd = a.to(dev), b.to(dev)
    c = torch.empty(M, N, device=dev)
    if mode == "NT":
        H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bd), K), H.mat(H._p(c), N))
    elif mode == "NN":
        bt = bd.t().contiguous()  # [K, N]
        H.gemm(H.AC_GEMM_NN, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bt), N), H.mat(H._p(c), N))
    else:
        at = ad.t().contiguous()  # [K, M]
        bt = bd.t().contiguous()
        if M % 4 or N % 4:
            pytest.skip("TN MFMA path needs M,N % 4 == 0 (falls to scalar kernel, covered below)")
        H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(at), M), H.mat(H._p(bt), N), H.mat(H._p(c), N))
    torch.cuda.synchronize()
    close(c, ref, name=f"gemm {mode}")


def test_gemm_simple_matches_mfma(dev):
    from applecider_amd import hipops as H
    M, N, K = 200, 136, 96
    a, b = g(dev, M, K, seed=3).to(dev), g(dev, N, K, seed=4).to(dev)
    c1, c2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c1), N))
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c2), N),
           force_simple=1)
    close(c1, c2, tol=1e-5, name="mfma vs scalar")
    # exact-integer check with asymmetric operands (catches transposed fragment maps)
    ai = torch.arange(M * K).reshape(M, K).remainder(7).float() - 3
    bi = torch.arange(N * K).reshape(N, K).remainder(5).float() - 2
    c3 = torch.empty(M, N, device=dev)
    ad, bd = ai.to(dev), bi.to(dev)
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bd), K), H.mat(H._p(c3), N))
    assert torch.equal(c3.cpu(), ai @ bi.t())


def test_gemm_epilogue_splitk_bf16(dev):
    from applecider_amd import hipops as H
    M, N, K = 512, 256, 2048
    a, b = g(dev, M, K, seed=5), g(dev, N, K, seed=6)
    bias, res, cs = g(dev, N, seed=7), g(dev, M, N, seed=8), g(dev, N, seed=9)
    ad, bd = a.to(dev), b.to(dev)
    c = torch.empty(M, N, device=dev)
    pre = torch.empty(M, N, device=dev)
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bd), K), H.mat(H._p(c), N),
           bias=bias.to(dev), act=H.ACT_GELU, pre_out=pre, ld_pre=N, colscale=cs.to(dev),
           residual=res.to(dev), ld_res=N)
    lin = a @ b.t() + bias
    close(pre, lin, name="pre_out")
    close(c, F.gelu(lin) * cs + res, name="epilogue")
    # split-K with atomics
    c2 = torch.zeros(M, N, device=dev)
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bd), K), H.mat(H._p(c2), N),
           accumulate=2, split_k=8)
    close(c2, a @ b.t(), name="split-k")
    # bf16 matrix-core path (inputs rounded to bf16)
    for mode in ("NT", "NN", "TN"):
        c3 = torch.empty(M, N, device=dev)
        if mode == "NT":
            H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bd), K),
                   H.mat(H._p(c3), N), math=1)
        elif mode == "NN":
            bt = bd.t().contiguous()
            H.gemm(H.AC_GEMM_NN, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bt), N),
                   H.mat(H._p(c3), N), math=1)
        else:
            at, bt = ad.t().contiguous(), bd.t().contiguous()
            H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(at), M), H.mat(H._p(bt), N),
                   H.mat(H._p(c3), N), math=1)
        ref = a.bfloat16().float() @ b.bfloat16().float().t()
        close(c3, ref, tol=1e-3, name=f"bf16 {mode}")


@pytest.mark.parametrize("act", [None, "gelu", "relu", "sigmoid", "tanh"])
@pytest.mark.parametrize("shape", [(3, 50, 96, 384), (1, 512, 288, 128), (1, 512, 2, 16), (1, 77, 19, 5)])
def test_linear_autograd(dev, act, shape):
    from applecider_amd import hipops as H
    lead, rows, K, N = shape
    x = g(dev, lead, rows, K, seed=1).requires_grad_()
    w = (g(dev, N, K, seed=2) / math.sqrt(K)).requires_grad_()
    b = g(dev, N, seed=3).requires_grad_()
    fn = {None: lambda t: t, "gelu": F.gelu, "relu": F.relu, "sigmoid": torch.sigmoid,
          "tanh": torch.tanh}[act]
    y = fn(F.linear(x, w, b))
    go = g(dev, *y.shape, seed=4)
    y.backward(go)
    xd, wd, bd = (t.detach().to(dev).requires_grad_() for t in (x, w, b))
    yd = H.linear(xd, wd, bd, act=act)
    yd.backward(go.to(dev))
    close(yd, y, name="y")
    close(xd.grad, x.grad, name="dx")
    close(wd.grad, w.grad, name="dw")
    close(bd.grad, b.grad, name="db")


def test_linear_residual_colscale(dev):
    from applecider_amd import hipops as H
    x = g(dev, 450, 384, seed=1).requires_grad_()
    w = (g(dev, 96, 384, seed=2) / 20).requires_grad_()
    b = g(dev, 96, seed=3).requires_grad_()
    gam = g(dev, 96, seed=4).requires_grad_()
    res = g(dev, 450, 96, seed=5).requires_grad_()
    y = res + gam * F.linear(x, w, b)
    go = g(dev, 450, 96, seed=6)
    y.backward(go)
    ts = [t.detach().to(dev).requires_grad_() for t in (x, w, b, gam, res)]
    yd = H.linear(ts[0], ts[1], ts[2], residual=ts[4], colscale=ts[3])
    yd.backward(go.to(dev))
    close(yd, y, name="y")
    for td, t, n in zip(ts, (x, w, b, gam, res), "x w b gamma res".split()):
        close(td.grad, t.grad, name="d" + n)


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,C", [(1000, 96), (333, 192), (64, 3072), (512, 5), (100, 130), (5000, 384),
                                    (700, 768), (300, 1536), (2000, 16), (9, 288), (4099, 64), (1237, 128),
                                    (801, 256), (77, 512), (3, 32), (20001, 64)])
@pytest.mark.parametrize("act", [None, "gelu"])
def test_layernorm(dev, rows, C, act):
    from applecider_amd import hipops as H
    x = (g(dev, rows, C, seed=1) * 2 + 0.5).requires_grad_()
    w = (1 + 0.1 * g(dev, C, seed=2)).requires_grad_()
    b = (0.1 * g(dev, C, seed=3)).requires_grad_()
    y = F.layer_norm(x, (C,), w, b, 1e-6)
    if act:
        y = F.gelu(y)
    go = g(dev, rows, C, seed=4)
    y.backward(go)
    xd, wd, bd = (t.detach().to(dev).requires_grad_() for t in (x, w, b))
    yd = H.layer_norm(xd, wd, bd, 1e-6, act=act)
    yd.backward(go.to(dev))
    close(yd, y, name="y")
    close(xd.grad, x.grad, name="dx")
    close(wd.grad, w.grad, name="dgamma")
    close(bd.grad, b.grad, name="dbeta")


@pytest.mark.parametrize("rows,C", [(1001, 64), (515, 96), (300, 128), (70, 384), (33, 1536)])
def test_layernorm_bf16_side_output(dev, bf16_mode, rows, C):
    """bf16 mode: LayerNorm leaves a bf16 copy of its output for the next matrix product; it must
    equal the rounding of the fp32 output, and Linear must pick it up (same result as a cast)."""
    from applecider_amd import hipops as H
    x = g(dev, rows, C, seed=1).to(dev)
    w, b = (1 + 0.1 * g(dev, C, seed=2)).to(dev), (0.1 * g(dev, C, seed=3)).to(dev)
    y = H.layer_norm(x, w, b, 1e-5, act="gelu")
    side = getattr(y, "_ac16", None)
    assert side is not None and side.dtype == torch.bfloat16
    assert torch.equal(side, y.bfloat16())
    wl = (g(dev, 64, C, seed=5) / 8).to(dev)
    out = H.linear(y, wl, None)
    y2 = y.clone()                      # no side output attached: the cast path
    close(out, H.linear(y2, wl, None), tol=0, name="linear from side output")


# ----------------------------------------------------------------------------- image branch
@pytest.mark.parametrize("B,H_,C", [(5, 15, 96), (3, 7, 192), (6, 3, 384), (9, 1, 768), (2, 15, 40)])
def test_dwconv(dev, B, H_, C):
    from applecider_amd import hipops as H
    x = g(dev, B, C, H_, H_, seed=1).requires_grad_()
    w = (g(dev, C, 1, 7, 7, seed=2) / 7).requires_grad_()
    b = g(dev, C, seed=3).requires_grad_()
    y = F.conv2d(x, w, b, padding=3, groups=C)
    go = g(dev, *y.shape, seed=4)
    y.backward(go)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()
    wd = w.detach().reshape(C, 49).t().contiguous().to(dev).requires_grad_()
    bd = b.detach().to(dev).requires_grad_()
    yd = H.dwconv7x7(xd, wd, bd)
    yd.backward(go.permute(0, 2, 3, 1).contiguous().to(dev))
    close(yd.permute(0, 3, 1, 2), y, name="y")
    close(xd.grad.permute(0, 3, 1, 2), x.grad, name="dx")
    close(wd.grad.t().reshape(C, 1, 7, 7), w.grad, name="dw")
    close(bd.grad, b.grad, name="db")


@pytest.mark.parametrize("B,H_,C", [(171, 15, 96), (513, 15, 96), (86, 15, 96), (128, 7, 192), (343, 7, 192), (700, 15, 32)])
def test_dwconv_pipelined_kernels(dev, B, H_, C):
    """The persistent LDS-DMA kernels (several (sample, slice) items per workgroup, double-buffered planes, counted
    vmcnt): batches that give workgroups unequal item counts, a single item, and more items than the grid — against
    torch's grouped conv2d (what timm's block dispatches to, astrominn.py:12-17) and bit-for-bit (forward, dx)
    against the one-item-per-workgroup kernels."""
    from applecider_amd import hipops as H
    assert B * (C // 32) >= 256, "shape would not take the pipelined path"
    x = g(dev, B, C, H_, H_, seed=1).requires_grad_()
    w = (g(dev, C, 1, 7, 7, seed=2) / 7).requires_grad_()
    b = g(dev, C, seed=3).requires_grad_()
    y = F.conv2d(x, w, b, padding=3, groups=C)
    go = g(dev, *y.shape, seed=4)
    y.backward(go)
    res = {}
    for variant in (0, 1):
        H._DWCONV_VARIANT = variant
        try:
            xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()
            wd = w.detach().reshape(C, 49).t().contiguous().to(dev).requires_grad_()
            bd = b.detach().to(dev).requires_grad_()
            yd = H.dwconv7x7(xd, wd, bd)
            yd.backward(go.permute(0, 2, 3, 1).contiguous().to(dev))
            torch.cuda.synchronize()
        finally:
            H._DWCONV_VARIANT = 0
        close(yd.permute(0, 3, 1, 2), y, name=f"y v{variant}")
        close(xd.grad.permute(0, 3, 1, 2), x.grad, name=f"dx v{variant}")
        close(wd.grad.t().reshape(C, 1, 7, 7), w.grad, name=f"dw v{variant}")
        close(bd.grad, b.grad, name=f"db v{variant}")
        res[variant] = (yd.detach(), xd.grad)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("B,H_,C,Co", [(4, 15, 96, 192), (4, 7, 192, 384), (8, 3, 384, 768), (3, 8, 32, 64)])
def test_patch_conv2x2(dev, B, H_, C, Co):
    from applecider_amd import hipops as H
    x = g(dev, B, C, H_, H_, seed=1).requires_grad_()
    w = (g(dev, Co, C, 2, 2, seed=2) / math.sqrt(4 * C)).requires_grad_()
    b = g(dev, Co, seed=3).requires_grad_()
    y = F.conv2d(x, w, b, stride=2)
    go = g(dev, *y.shape, seed=4)
    y.backward(go)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()
    wd = w.detach().permute(0, 2, 3, 1).reshape(Co, 4 * C).contiguous().to(dev).requires_grad_()
    bd = b.detach().to(dev).requires_grad_()
    yd = H.patch_conv2x2(xd, wd, bd)
    yd.backward(go.permute(0, 2, 3, 1).contiguous().to(dev))
    close(yd.permute(0, 3, 1, 2), y, name="y")
    close(xd.grad.permute(0, 3, 1, 2), x.grad, name="dx")
    close(wd.grad.reshape(Co, 2, 2, C).permute(0, 3, 1, 2), w.grad, name="dw")
    close(bd.grad, b.grad, name="db")


def test_stem(dev):
    from applecider_amd import hipops as H
    B = 6
    img = g(dev, B, 3, 63, 63, seed=1)
    w = (g(dev, 96, 3, 4, 4, seed=2) / 7).requires_grad_()
    b = g(dev, 96, seed=3).requires_grad_()
    y = F.conv2d(img, w, b, stride=4)  # [B,96,15,15]
    go = g(dev, *y.shape, seed=4)
    y.backward(go)
    w64 = torch.zeros(96, 64)
    w64[:, :48] = w.detach().permute(0, 2, 3, 1).reshape(96, 48)
    wd = w64.to(dev).requires_grad_()
    bd = b.detach().to(dev).requires_grad_()
    patches, OH, OW = H.stem_patchify(img.to(dev))
    assert (OH, OW) == (15, 15)
    yd = H.linear(patches, wd, bd)
    yd.backward(go.permute(0, 2, 3, 1).reshape(-1, 96).contiguous().to(dev))
    close(yd.reshape(B, 15, 15, 96).permute(0, 3, 1, 2), y, name="y")
    close(wd.grad[:, :48].reshape(96, 4, 4, 3).permute(0, 3, 1, 2), w.grad, name="dw")
    assert wd.grad[:, 48:].abs().max().item() == 0.0
    close(bd.grad, b.grad, name="db")


def test_pools(dev):
    from applecider_amd import hipops as H
    x = g(dev, 3, 64, 40, seed=1).requires_grad_()  # [B, L, C]
    y = F.max_pool1d(x.permute(0, 2, 1), 4).permute(0, 2, 1)
    go = g(dev, *y.shape, seed=2)
    y.backward(go)
    xd = x.detach().to(dev).requires_grad_()
    yd = H.maxpool4(xd)
    yd.backward(go.to(dev))
    close(yd, y, tol=0, name="maxpool")
    close(xd.grad, x.grad, tol=0, name="maxpool dx")
    x.grad = None
    # shapes off the 16-byte path (C % 4 != 0, L % 4 != 0: tail rows get a zero gradient) and ties
    for shape in ((2, 66, 41), (2, 64, 6), (5, 128, 64)):
        xs = g(dev, *shape, seed=7).round().requires_grad_()        # rounded: many equal maxima
        ys = F.max_pool1d(xs.permute(0, 2, 1), 4).permute(0, 2, 1)
        gs = g(dev, *ys.shape, seed=8)
        ys.backward(gs)
        xsd = xs.detach().to(dev).requires_grad_()
        ysd = H.maxpool4(xsd)
        ysd.backward(gs.to(dev))
        close(ysd, ys, tol=0, name=f"maxpool {shape}")
        close(xsd.grad, xs.grad, tol=0, name=f"maxpool dx {shape}")
    y = F.adaptive_max_pool1d(x.permute(0, 2, 1), 1).squeeze(-1)
    go = g(dev, *y.shape, seed=3)
    y.backward(go)
    xd = x.detach().to(dev).requires_grad_()
    yd = H.global_max(xd)
    yd.backward(go.to(dev))
    close(yd, y, tol=0, name="globalmax")
    close(xd.grad, x.grad, tol=0, name="globalmax dx")
    x.grad = None
    y = x.mean(1)
    y.backward(go)
    xd = x.detach().to(dev).requires_grad_()
    yd = H.avgpool_tokens(xd)
    yd.backward(go.to(dev))
    close(yd, y, name="avgpool")
    close(xd.grad, x.grad, name="avgpool dx")


# ----------------------------------------------------------------------------- spectra conv bank
@pytest.mark.parametrize("fused_ln", [False, True])
@pytest.mark.parametrize("B,L,Cin,Cout,ks", [(2, 256, 1, 64, (3, 61, 1021)), (3, 64, 1, 32, (3, 7, 13)),
                                             (2, 128, 64, 128, (3, 31, 251)), (2, 16, 32, 64, (3, 7, 13)),
                                             (1, 64, 128, 32, (3, 15, 61)), (4, 64, 256, 512, (3, 11, 31)),
                                             (2, 16, 512, 1024, (3, 7, 13))])
def test_conv_group1d(dev, B, L, Cin, Cout, ks, fused_ln):
    from applecider_amd import hipops as H
    x = g(dev, B, Cin, L, seed=1).requires_grad_(Cin != 1)
    ws = [(g(dev, Cout, Cin, k, seed=10 + i) / math.sqrt(Cin * k)).requires_grad_() for i, k in enumerate(ks)]
    bs = [g(dev, Cout, seed=20 + i).requires_grad_() for i in range(len(ks))]
    lw = (1 + 0.1 * g(dev, 3 * Cout, seed=30)).requires_grad_()
    lb = (0.1 * g(dev, 3 * Cout, seed=31)).requires_grad_()
    y = torch.cat([F.conv1d(x, w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)  # [B, 3Cout, L]
    if fused_ln:
        y = F.gelu(F.layer_norm(y.permute(0, 2, 1), (3 * Cout,), lw, lb, 1e-5)).permute(0, 2, 1)
    go = g(dev, *y.shape, seed=5)
    y.backward(go)
    xd = x.detach().permute(0, 2, 1).contiguous().to(dev).requires_grad_(Cin != 1)
    wd = [w.detach().permute(0, 2, 1).reshape(Cout, -1).contiguous().to(dev).requires_grad_() for w in ws]
    bd = [b.detach().to(dev).requires_grad_() for b in bs]
    lwd, lbd = lw.detach().to(dev).requires_grad_(), lb.detach().to(dev).requires_grad_()
    yd = H.conv_group1d(xd, ks, wd, bd, ln=(lwd, lbd, 1e-5) if fused_ln else None)
    yd.backward(go.permute(0, 2, 1).contiguous().to(dev))
    close(yd.permute(0, 2, 1), y, name="y")
    for i, k in enumerate(ks):
        close(wd[i].grad.reshape(Cout, k, Cin).permute(0, 2, 1), ws[i].grad, name=f"dw{i}")
        close(bd[i].grad, bs[i].grad, name=f"db{i}")
    if Cin != 1:
        close(xd.grad.permute(0, 2, 1), x.grad, name="dx")
    if fused_ln:
        close(lwd.grad, lw.grad, name="dgamma")
        close(lbd.grad, lb.grad, name="dbeta")


# ----------------------------------------------------------------------------- photometry branch
def test_embed(dev):
    from applecider_amd import hipops as H
    B, L, D = 3, 20, 128
    x = g(dev, B, L, 7, seed=1)
    W = (g(dev, D, 7, seed=2) / 3).requires_grad_()
    bias = g(dev, D, seed=3).requires_grad_()
    w0, b0 = g(dev, 1, seed=4).requires_grad_(), g(dev, 1, seed=5).requires_grad_()
    w, b = g(dev, D - 1, seed=6).requires_grad_(), g(dev, D - 1, seed=7).requires_grad_()
    cls = g(dev, 1, 1, D, seed=8).requires_grad_()
    t = x[..., 0]
    te = torch.cat([(w0 * t + b0).unsqueeze(-1), torch.sin(t.unsqueeze(-1) * w + b)], -1)
    h = torch.cat([cls.expand(B, -1, -1), F.linear(x, W, bias) + te], 1)
    go = g(dev, *h.shape, seed=9)
    h.backward(go)
    x8 = H.pad_channels(x.to(dev), 8)
    W8 = torch.zeros(D, 8)
    W8[:, :7] = W.detach()
    W8d = W8.to(dev).requires_grad_()
    biasd = bias.detach().to(dev).requires_grad_()
    twd = torch.cat([w0.detach(), w.detach()]).to(dev).requires_grad_()
    tbd = torch.cat([b0.detach(), b.detach()]).to(dev).requires_grad_()
    clsd = cls.detach().reshape(D).to(dev).requires_grad_()
    hd = H.embed(x8, W8d, biasd, twd, tbd, clsd)
    hd.backward(go.to(dev))
    close(hd, h, name="h")
    close(W8d.grad[:, :7], W.grad, name="dW")
    close(biasd.grad, bias.grad, name="dbias")
    close(twd.grad, torch.cat([w0.grad, w.grad]), name="dtw")
    close(tbd.grad, torch.cat([b0.grad, b.grad]), name="dtb")
    close(clsd.grad, cls.grad.reshape(D), name="dcls")


@pytest.mark.parametrize("B,T,H_", [(3, 129, 8), (2, 258, 8), (4, 20, 4)])
def test_mha(dev, B, T, H_):
    from applecider_amd import hipops as H
    Dh = 16
    D = H_ * Dh
    qkv = g(dev, B, T, 3 * D, seed=1).requires_grad_()
    lens = [T, max(2, T // 3), 1, 7][:B]
    pad = torch.zeros(B, T, dtype=torch.bool)
    for i, n in enumerate(lens):
        pad[i, n:] = True
    q, k, v = qkv.split(D, -1)
    sh = lambda t: t.reshape(B, T, H_, Dh).permute(0, 2, 1, 3)
    s = (sh(q) / math.sqrt(Dh)) @ sh(k).transpose(-1, -2)
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, -1) @ sh(v)).permute(0, 2, 1, 3).reshape(B, T, D)
    go = g(dev, B, T, D, seed=2)
    o.backward(go)
    qd = qkv.detach().to(dev).requires_grad_()
    od = H.mha(qd, pad.to(torch.uint8).to(dev), H_)
    od.backward(go.to(dev))
    close(od, o, name="out")
    close(qd.grad, qkv.grad, name="dqkv")


def test_mha_dropout_consistency(dev):
    """Dropout inside attention: forward/backward must use the same mask -> check the gradient of
    a linear functional by finite differences along a random direction."""
    from applecider_amd import hipops as H
    B, T, H_, D = 2, 33, 8, 128
    qkv = g(dev, B, T, 3 * D, seed=1).to(dev)
    go = g(dev, B, T, D, seed=2).to(dev)
    dirn = g(dev, B, T, 3 * D, seed=3).to(dev)
    pad = torch.zeros(B, T, dtype=torch.uint8, device=dev)
    seed = 1234567
    f = lambda t: (H._MHA.apply(t, pad, H_, 0.4, seed) * go).sum()
    q = qkv.clone().requires_grad_()
    f(q).backward()
    analytic = (q.grad * dirn).sum().item()
    e = 1e-2
    numeric = (f(qkv + e * dirn).item() - f(qkv - e * dirn).item()) / (2 * e)
    assert abs(analytic - numeric) <= 2e-2 * max(1.0, abs(numeric)), (analytic, numeric)
    # keep-rate statistics through V = ones: each output is sum_j keep_ij P_ij / (1-p), mean 1
    qkv1 = qkv.clone()
    qkv1[..., 2 * D:] = 1.0
    o = H._MHA.apply(qkv1, pad, H_, 0.4, seed)
    assert abs(o.mean().item() - 1.0) < 0.02


# ----------------------------------------------------------------------------- small ops
def test_small_ops(dev):
    from applecider_amd import hipops as H
    a, b_, s = (g(dev, 100, 32, seed=i).requires_grad_() for i in (1, 2, 3))
    y = a * b_ + s
    go = g(dev, 100, 32, seed=4)
    y.backward(go)
    ad, bd, sd = (t.detach().to(dev).requires_grad_() for t in (a, b_, s))
    yd = H.gate(ad, bd, sd)
    yd.backward(go.to(dev))
    close(yd, y, name="gate")
    close(ad.grad, a.grad), close(bd.grad, b_.grad), close(sd.grad, s.grad)
    # cat / gather / l2norm / add / act / take_token
    md = g(dev, 50, 24, seed=5)
    idx = [6, 9, 10, 13, 15, 17, 18]
    close(H.gather_cols(md.to(dev), torch.tensor(idx, dtype=torch.int32, device=dev)), md[:, idx], tol=0)
    parts = [g(dev, 50, n, seed=10 + n).requires_grad_() for n in (32, 5, 32)]
    yc = torch.cat(parts, 1)
    goc = g(dev, *yc.shape, seed=6)
    yc.backward(goc)
    pd_ = [p.detach().to(dev).requires_grad_() for p in parts]
    ycd = H.cat_cols(pd_)
    ycd.backward(goc.to(dev))
    close(ycd, yc, tol=0)
    for p, q in zip(pd_, parts):
        close(p.grad, q.grad, tol=0)
    x = g(dev, 64, 64, seed=7).requires_grad_()
    yn = x / x.norm(dim=-1, keepdim=True)
    gon = g(dev, 64, 64, seed=8)
    yn.backward(gon)
    xd = x.detach().to(dev).requires_grad_()
    ynd = H.l2_normalize(xd)
    ynd.backward(gon.to(dev))
    close(ynd, yn), close(xd.grad, x.grad, name="l2 dx")
    for kind, fn in (("gelu", F.gelu), ("tanh", torch.tanh), ("sigmoid", torch.sigmoid), ("relu", F.relu)):
        x.grad = None
        ya = fn(x)
        ya.backward(gon)
        xd = x.detach().to(dev).requires_grad_()
        yad = H.activation(xd, kind)
        yad.backward(gon.to(dev))
        close(yad, ya, name=kind), close(xd.grad, x.grad, name="d" + kind)
    z = g(dev, 4, 9, 16, seed=9).requires_grad_()
    z[:, 0].backward(g(dev, 4, 16, seed=10))
    zd = z.detach().to(dev).requires_grad_()
    td = H.take_token(zd, 0)
    td.backward(g(dev, 4, 16, seed=10).to(dev))
    close(td, z[:, 0], tol=0), close(zd.grad, z.grad, tol=0)
    close(H.add(ad.detach(), bd.detach(), 1 / 3), (a + b_) / 3)
    close(H.softmax_rows(md.to(dev)), torch.softmax(md, -1))


def test_moe_top2(dev):
    from applecider_amd import hipops as H
    B, E, Cn = 200, 4, 5
    sc = torch.sigmoid(g(dev, B, E, seed=1)).requires_grad_()
    eo = g(dev, E, B, Cn, seed=2).requires_grad_()
    tw, ti = torch.topk(sc, k=2, dim=-1)
    out = torch.zeros(B, Cn)
    for e in range(E):
        mask = (ti == e).any(-1)
        if mask.any():
            wts = tw[mask, (ti[mask] == e).nonzero()[:, 1]]
            out[mask] = out[mask] + wts.unsqueeze(-1) * eo[e][mask]
    go = g(dev, B, Cn, seed=3)
    out.backward(go)
    scd, eod = sc.detach().to(dev).requires_grad_(), eo.detach().to(dev).requires_grad_()
    outd, sel = H.moe_top2(scd, eod)
    outd.backward(go.to(dev))
    close(outd, out, tol=1e-6)
    assert torch.equal(sel.cpu().long(), ti)
    close(scd.grad, sc.grad, tol=1e-6), close(eod.grad, eo.grad, tol=1e-6)


def test_losses(dev):
    from applecider_amd import hipops as H
    B, Cn = 300, 5
    z = (g(dev, B, Cn, seed=1) * 2).requires_grad_()
    t = torch.randint(0, Cn, (B,), generator=torch.Generator().manual_seed(2))
    onehot = F.one_hot(t, Cn).float()

    def focal(logits, target, gamma, alpha, eps):
        logp = F.log_softmax(logits, 1)
        p = logp.exp()
        if eps > 0:
            y = torch.full_like(logp, eps / (Cn - 1))
            y.scatter_(1, target.unsqueeze(1), 1 - eps)
        else:
            y = F.one_hot(target, Cn).float()
        fw = (1 - p).pow(gamma)
        if alpha is not None:
            fw = fw * alpha.view(1, Cn)
        return -(y * fw * logp).sum(1).mean()

    alpha = torch.tensor([0.3, 0.1, 0.1, 0.3, 0.2])
    cases = [("soft", lambda zz: F.cross_entropy(zz, onehot), lambda zd: H.cross_entropy_soft(zd, onehot.to(dev))),
             ("index", lambda zz: F.cross_entropy(zz, t), lambda zd: H.cross_entropy_index(zd, t.to(dev)))]
    for gamma in (0.0, 2.0, 1.5):
        for al in (None, alpha):
            for eps in (0.0, 0.1):
                cases.append((f"focal g{gamma} a{al is not None} e{eps}",
                              lambda zz, gamma=gamma, al=al, eps=eps: focal(zz, t, gamma, al, eps),
                              lambda zd, gamma=gamma, al=al, eps=eps: H.focal_loss(
                                  zd, t.to(dev), gamma, al.to(dev) if al is not None else None, eps)))
    for name, ref_fn, hip_fn in cases:
        z.grad = None
        l = ref_fn(z)
        l.backward()
        zd = z.detach().to(dev).requires_grad_()
        ld = hip_fn(zd)
        ld.backward()
        assert abs(ld.item() - l.item()) <= 1e-5 * max(1, abs(l.item())), name
        close(zd.grad, z.grad, tol=1e-4, name=name)


def test_dropout_stats(dev):
    from applecider_amd import hipops as H
    x = torch.ones(1 << 20, device=dev, requires_grad=True)
    for p in (0.25, 0.4, 0.5):
        y = H.dropout(x, p, True)
        keep = (y != 0).float().mean().item()
        assert abs(keep - (1 - p)) < 5e-3
        assert abs(y.mean().item() - 1.0) < 1e-2
        x.grad = None
        y.sum().backward()
        assert torch.equal(x.grad, y.detach())  # same mask, same scale
    y1, y2 = H.dropout(x, 0.5, True), H.dropout(x, 0.5, True)
    assert not torch.equal(y1, y2)  # fresh seed per call
    assert H.dropout(x, 0.5, False) is x
    # the counter hash must not show structure along rows, columns or between consecutive seeds
    k1, k2 = (y1 != 0).float().reshape(1024, 1024), (y2 != 0).float().reshape(1024, 1024)
    assert (k1.mean(0) - 0.5).abs().max().item() < 0.08 and (k1.mean(1) - 0.5).abs().max().item() < 0.08
    for a, b in ((k1[:, 1:], k1[:, :-1]), (k1[1:], k1[:-1]), (k1, k2)):
        corr = ((a - 0.5) * (b - 0.5)).mean().item() * 4
        assert abs(corr) < 6e-3, corr


def test_optimizers(dev):
    from applecider_amd import hipops as H
    from applecider_amd._lib import AdamSeg
    n = 10000
    p0, g0 = g(dev, n, seed=1), g(dev, n, seed=2)
    for decoupled, Opt in ((1, torch.optim.AdamW), (0, torch.optim.Adam)):
        p = p0.clone().requires_grad_()
        opt = Opt([{"params": [p], "lr": 3e-3, "weight_decay": 0.05, "betas": (0.9, 0.99)}], eps=5e-10)
        pd_, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        for step in range(1, 4):
            gr = g0 * step
            p.grad = gr.clone()
            opt.step()
            H.adam_flat(pd_, gr.to(dev), m, v, [AdamSeg(0, n // 2, 3e-3, 0.9, 0.99, 5e-10, 0.05, decoupled),
                                                AdamSeg(n // 2, n, 3e-3, 0.9, 0.99, 5e-10, 0.05, decoupled)], step)
        close(pd_, p, tol=1e-5, name=f"adam decoupled={decoupled}")
    p = p0.clone().requires_grad_()
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9)
    pd_, buf = p0.clone().to(dev), torch.zeros(n, device=dev)
    for step in range(3):
        p.grad = g0.clone()
        opt.step()
        H.sgd_flat(pd_, g0.to(dev), buf, 0.01, 0.9, 0.0, step == 0)
    close(pd_, p, tol=1e-6, name="sgd")
    gd = (g0 * 3).to(dev)
    coef, ss = H.clip_coef(gd, 1.0)
    tot = (g0 * 3).norm().item()
    assert abs(math.sqrt(ss.item()) - tot) < 1e-3 * tot
    assert abs(coef.item() - min(1.0, 1.0 / (tot + 1e-6))) < 1e-6


# ----------------------------------------------------------------------------- bf16-operand path
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 136, 520), (128, 128, 16064), (304, 64, 96)])
def test_gemm_bf16_operands(dev, M, N, K):
    """NT and TN kernels that read bf16 operands (ds_read_b128 / ds_read_b64_tr_b16 fragments)
    against an fp32 product of the same bf16-rounded inputs (exact up to summation order)."""
    from applecider_amd import hipops as H
    a, b = g(dev, M, K, seed=1), g(dev, N, K, seed=2)
    ar, br = a.bfloat16().float(), b.bfloat16().float()
    ref = ar @ br.t()
    ad, bd = a.to(dev), b.to(dev)
    a16, b16 = H.cast16(ad), H.cast16(bd)
    assert torch.equal(a16.float().cpu(), ar)
    c = torch.empty(M, N, device=dev)
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a16), K), H.mat(H._p(b16), K), H.mat(H._p(c), N), math=2)
    close(c, ref, tol=2e-5, name="bf16in NT")
    # exact integers through the transposed-read path: C = A^T B with A [K, M], B [K, N]
    at16, bt16 = H.cast16_T(ad), H.cast16_T(bd)  # [K, M], [K, N]
    assert torch.equal(at16.float().cpu(), ar.t())
    c2 = torch.empty(M, N, device=dev)
    H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(at16), M), H.mat(H._p(bt16), N), H.mat(H._p(c2), N), math=2)
    close(c2, ref, tol=2e-5, name="bf16in TN")
    ai = (torch.arange(M * K).reshape(M, K).remainder(7).float() - 3)
    bi = (torch.arange(N * K).reshape(N, K).remainder(5).float() - 2)
    c3 = torch.zeros(M, N, device=dev)
    ait, bit = H.cast16_T(ai.to(dev)), H.cast16_T(bi.to(dev))  # keep the operands alive
    H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(ait), M), H.mat(H._p(bit), N), H.mat(H._p(c3), N),
           math=2, accumulate=2, split_k=3)
    assert torch.equal(c3.cpu(), ai @ bi.t())


@pytest.mark.parametrize("tile", [3, 4])
@pytest.mark.parametrize("M,N,K,split", [(128, 256, 640, 1), (200, 712, 520, 3), (384, 1032, 2048, 4)])
def test_gemm_bf16_tn_wide_tiles(dev, tile, M, N, K, split):
    """The 8-wave TN tiles (256 x 128 and 128 x 256, the latter used for the long Conv1d weight
    gradients): exact integers, ragged edges, split-K atomics."""
    from applecider_amd import hipops as H
    ai = (torch.arange(M * K).reshape(M, K).remainder(7).float() - 3)
    bi = (torch.arange(N * K).reshape(N, K).remainder(5).float() - 2)
    c = torch.zeros(M, N, device=dev)
    ait, bit = H.cast16_T(ai.to(dev)), H.cast16_T(bi.to(dev))   # [K, M], [K, N]
    H.gemm(H.AC_GEMM_TN, M, N, K, H.mat(H._p(ait), M), H.mat(H._p(bit), N), H.mat(H._p(c), N),
           math=2, accumulate=2, split_k=split, tile=tile)
    assert torch.equal(c.cpu(), ai @ bi.t())
    if tile == 3:   # the same tile shape on the NT side (long, wide products pick it automatically)
        c_nt = torch.empty(M, N, device=dev)
        a16, b16 = H.cast16(ai.to(dev)), H.cast16(bi.to(dev))
        H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a16), K), H.mat(H._p(b16), K), H.mat(H._p(c_nt), N),
               math=2, tile=3)
        assert torch.equal(c_nt.cpu(), ai @ bi.t())
    # Toeplitz view as the B operand (the conv weight gradient): rows overlap by Cin elements
    B_, L, Cin, k = 3, 64, 8, 32
    Lp = L + k - 1
    x = torch.randint(-3, 4, (B_, Lp, Cin)).float()
    dy = torch.randint(-2, 3, (B_ * L, M)).float()
    x16, dy16 = H.cast16(x.to(dev)), H.cast16(dy.to(dev))
    dw = torch.zeros(M, k * Cin, device=dev)
    H.gemm(H.AC_GEMM_TN, M, k * Cin, B_ * L, H.mat(H._p(dy16), M),
           H.mat(H._p(x16), r1=L, r2=L, s1=Lp * Cin, s3=Cin), H.mat(H._p(dw), k * Cin),
           math=2, accumulate=2, split_k=2, tile=tile)
    win = x.unfold(1, k, 1).permute(0, 1, 3, 2).reshape(B_ * L, k * Cin)   # [B*L, (tap, ci)]
    assert torch.equal(dw.cpu(), dy.t() @ win)


@pytest.fixture
def bf16_mode():
    from applecider_amd import hipops as H
    H.set_math("bf16")
    yield
    H.set_math("f32")


@pytest.mark.parametrize("B,L,Cin,Cout,ks", [(2, 256, 1, 64, (3, 61, 1021)), (2, 128, 64, 128, (3, 31, 251)),
                                             (4, 64, 256, 512, (3, 11, 31)), (3, 512, 64, 128, (3, 31, 251)),
                                             (2, 256, 128, 256, (3, 15, 61)), (2, 128, 64, 64, (5, 9, 33)),
                                             (1, 256, 256, 256, (3, 11, 31))])
@pytest.mark.parametrize("window", [True, False])
def test_conv_group1d_bf16(dev, bf16_mode, B, L, Cin, Cout, ks, window):
    """bf16 matrix-core mode of the conv bank: same math as fp32 with operands rounded to bf16
    (fp32 accumulate), so it is compared with torch on bf16-rounded x / w / dy at 3e-3."""
    from applecider_amd import hipops as H
    if window and (Cin == 1 or L % 128):
        pytest.skip("LDS-window kernel does not cover this shape")
    H.enable_conv_window(window)
    rb = lambda t: t.bfloat16().float()
    x = rb(g(dev, B, Cin, L, seed=1)).requires_grad_(Cin != 1)
    ws = [rb(g(dev, Cout, Cin, k, seed=10 + i) / math.sqrt(Cin * k)).requires_grad_() for i, k in enumerate(ks)]
    bs = [g(dev, Cout, seed=20 + i).requires_grad_() for i in range(len(ks))]
    y = torch.cat([F.conv1d(x, w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)
    go = rb(g(dev, *y.shape, seed=5))
    y.backward(go)
    xd = x.detach().permute(0, 2, 1).contiguous().to(dev).requires_grad_(Cin != 1)
    wd = [w.detach().permute(0, 2, 1).reshape(Cout, -1).contiguous().to(dev).requires_grad_() for w in ws]
    bd = [b.detach().to(dev).requires_grad_() for b in bs]
    try:
        yd = H.conv_group1d(xd, ks, wd, bd)
        yd.backward(go.permute(0, 2, 1).contiguous().to(dev))
    finally:
        H.enable_conv_window(True)
    close(yd.permute(0, 2, 1), y, tol=1e-4, name="y")
    for i, k in enumerate(ks):
        close(wd[i].grad.reshape(Cout, k, Cin).permute(0, 2, 1), ws[i].grad, tol=1e-4, name=f"dw{i}")
    if Cin != 1:
        close(xd.grad.permute(0, 2, 1), x.grad, tol=1e-4, name="dx")


def test_linear_and_patchconv_bf16(dev, bf16_mode):
    from applecider_amd import hipops as H
    rb = lambda t: t.bfloat16().float()
    x = rb(g(dev, 900, 384, seed=1)).requires_grad_()
    w = rb(g(dev, 96, 384, seed=2) / 20).requires_grad_()
    b = g(dev, 96, seed=3).requires_grad_()
    y = F.linear(x, w, b)
    go = rb(g(dev, *y.shape, seed=4))
    y.backward(go)
    xd, wd, bd = (t.detach().to(dev).requires_grad_() for t in (x, w, b))
    yd = H.linear(xd, wd, bd)
    yd.backward(go.to(dev))
    close(yd, y, tol=1e-4), close(xd.grad, x.grad, tol=1e-4), close(wd.grad, w.grad, tol=1e-4)
    close(bd.grad, b.grad, tol=1e-4)
    B, H_, C, Co = 4, 7, 192, 384
    x = rb(g(dev, B, C, H_, H_, seed=1)).requires_grad_()
    w = rb(g(dev, Co, C, 2, 2, seed=2) / math.sqrt(4 * C)).requires_grad_()
    y = F.conv2d(x, w, None, stride=2)
    go = rb(g(dev, *y.shape, seed=4))
    y.backward(go)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()
    wd = w.detach().permute(0, 2, 3, 1).reshape(Co, 4 * C).contiguous().to(dev).requires_grad_()
    yd = H.patch_conv2x2(xd, wd, None)
    yd.backward(go.permute(0, 2, 3, 1).contiguous().to(dev))
    close(yd.permute(0, 3, 1, 2), y, tol=1e-4, name="y")
    close(xd.grad.permute(0, 3, 1, 2), x.grad, tol=1e-4, name="dx")
    close(wd.grad.reshape(Co, 2, 2, C).permute(0, 3, 1, 2), w.grad, tol=1e-4, name="dw")


# ----------------------------------------------------------------------------- fused bf16-hidden MLP
def test_mlp_bf16_gelu_layerscale(dev, bf16_mode):
    """ConvNeXt block MLP (GELU, layer scale, residual) with the hidden kept in bf16 only, against a
    torch restatement that rounds at the same points (operands, hidden activation, hidden gradient)."""
    from applecider_amd import hipops as H
    rb = lambda t: t.bfloat16().float()
    M, Cn, Hd = 900, 96, 384
    x = rb(g(dev, M, Cn, seed=1))
    w1, b1 = rb(g(dev, Hd, Cn, seed=2) / 10), g(dev, Hd, seed=3)
    w2, b2 = rb(g(dev, Cn, Hd, seed=4) / 20), g(dev, Cn, seed=5)
    gamma, res = g(dev, Cn, seed=6), g(dev, M, Cn, seed=7)
    dy = g(dev, M, Cn, seed=8)
    h = x @ w1.t() + b1
    g16 = rb(F.gelu(h))
    ylin = g16 @ w2.t() + b2
    y = ylin * gamma + res
    g2 = dy * gamma
    g2_16 = rb(g2)
    hh = h.clone().requires_grad_()
    F.gelu(hh).sum().backward()
    dh = rb((g2_16 @ w2) * hh.grad)
    ref = {"dx": dh @ w1, "dw1": dh.t() @ x, "db1": dh.sum(0), "dw2": g2_16.t() @ g16, "db2": g2.sum(0),
           "dgamma": (dy * ylin).sum(0)}
    t = lambda a: a.detach().to(dev).requires_grad_()
    xd, w1d, b1d, w2d, b2d, gd, rd = (t(a) for a in (x, w1, b1, w2, b2, gamma, res))
    yd = H.mlp(xd, w1d, b1d, w2d, b2d, "gelu", residual=rd, colscale=gd)
    yd.backward(dy.to(dev))
    close(yd, y, tol=1e-3, name="y")          # a near-tie bf16 rounding of one hidden value may flip
    close(rd.grad, dy, tol=0, name="dres")
    for name, got in (("dx", xd.grad), ("dw1", w1d.grad), ("db1", b1d.grad), ("dw2", w2d.grad),
                      ("db2", b2d.grad), ("dgamma", gd.grad)):
        close(got, ref[name], tol=3e-3, name=name)


def test_mlp_bf16_relu_dropout_matches_unfused(dev, bf16_mode):
    """Encoder feed-forward with both dropouts fused into the GEMM epilogues against the unfused
    kernels (Linear, Dropout, Add) drawing the same seeds: same masks, same bf16 roundings."""
    import itertools
    from applecider_amd import hipops as H
    M, Cn, Hd, p = 1290, 128, 512, 0.4
    x, res_same = g(dev, M, Cn, seed=1), True
    w1, b1 = g(dev, Hd, Cn, seed=2) / 10, g(dev, Hd, seed=3)
    w2, b2 = g(dev, Cn, Hd, seed=4) / 20, g(dev, Cn, seed=5)
    dy = g(dev, M, Cn, seed=8).to(dev)
    t = lambda a: a.detach().to(dev).requires_grad_()

    def run(fused):
        H._seed_counter = itertools.count(5000)
        xd, w1d, b1d, w2d, b2d = (t(a) for a in (x, w1, b1, w2, b2))
        if fused:
            y = H.mlp(xd, w1d, b1d, w2d, b2d, "relu", p1=p, p2=p, training=True, residual=xd)
        else:
            hd = H.dropout(H.linear(xd, w1d, b1d, act="relu"), p, True)
            y = H.add(xd, H.dropout(H.linear(hd, w2d, b2d), p, True))
        y.backward(dy)
        return [y.detach()] + [a.grad for a in (xd, w1d, b1d, w2d, b2d)]

    a, b = run(True), run(False)
    zeros = float((a[0] - x.to(dev) == 0).float().mean())
    assert 0.3 < zeros < 0.5, zeros                     # dropout2 really dropped ~p of the update
    for name, u, v, tol in zip(("y", "dx", "dw1", "db1", "dw2", "db2"), a, b,
                               (1e-5, 5e-4, 5e-4, 5e-3, 1e-5, 1e-5)):  # db1: summed from the bf16 hidden gradient
        close(u, v, tol=tol, name=name)
    # eval mode: no dropout, still the fused path
    xd, w1d, b1d, w2d, b2d = (t(a_) for a_ in (x, w1, b1, w2, b2))
    ye = H.mlp(xd, w1d, b1d, w2d, b2d, "relu", p1=p, p2=p, training=False, residual=xd)
    rb = lambda t_: t_.bfloat16().float()
    ref = x + rb(F.relu(rb(x) @ rb(w1).t() + b1)) @ rb(w2).t() + b2
    close(ye, ref, tol=5e-4, name="eval")   # near-tie bf16 roundings of the hidden may flip


def test_colsum_bf16(dev):
    from applecider_amd import hipops as H
    x = g(dev, 1000, 386, seed=3).bfloat16().to(dev)
    out = torch.zeros(386, device=dev)
    H._lib.check(H._lib_().ac_colsum_bf16(H._p(x), 386, H._p(out), 1000, 386, 0, H._stream()), "colsum")
    close(out, x.float().sum(0), tol=1e-5, name="colsum_bf16")


def test_bf16_parameter_mirrors(dev, bf16_mode):
    """Weights flattened by the optimizer are cast (and transposed) for all layers in two launches per
    step; the views must track optimizer steps, load_state_dict-style writes and plain tensors."""
    from applecider_amd import hipops as H
    from applecider_amd.optim import FlatAdam
    lin = torch.nn.Linear(72, 40).to(dev)
    other = torch.nn.Linear(16, 24).to(dev)
    opt = FlatAdam([{"params": list(lin.parameters()) + list(other.parameters())}], lr=0.1).prepare()
    w = lin.weight
    assert torch.equal(H.cast16_w(w), w.detach().bfloat16())
    assert torch.equal(H.cast16_wT(w), w.detach().t().contiguous().bfloat16())
    assert H.cast16_w(w).data_ptr() == opt.fp.flat16.data_ptr() + 2 * opt.fp.offsets[0]   # a view
    assert torch.equal(H.cast16_wT(other.weight), other.weight.detach().t().contiguous().bfloat16())
    opt.zero_grad()
    x = torch.randn(64, 72, device=dev)
    H.linear(x, lin.weight, lin.bias).sum().backward()
    before = w.detach().clone()
    opt.step()
    assert not torch.equal(before, w.detach())
    assert torch.equal(H.cast16_w(w), w.detach().bfloat16())          # refreshed after the step
    with torch.no_grad():
        w.copy_(torch.randn_like(w))                                    # checkpoint-style write
    assert torch.equal(H.cast16_wT(w), w.detach().t().contiguous().bfloat16())
    plain = torch.randn(8, 16, device=dev)                             # not a flattened parameter
    assert torch.equal(H.cast16_w(plain), plain.bfloat16())


def test_conv_bank_bf16_handover(dev, bf16_mode):
    """Pooled SpectraNet block in bf16 mode: LN+GELU output and the gradient coming back cross the
    conv bank / 1x1 conv boundary in bf16 only (fp32 placeholders are never written).  Same forward as
    the fp32 hand-over; gradients differ by one bf16 rounding of d(LN output)."""
    from applecider_amd import hipops as H
    B, L, Cin, Cout, ks = 3, 256, 64, 64, (3, 15, 31)
    x = g(dev, B, L, Cin, seed=1)
    ws = [g(dev, Cout, k * Cin, seed=10 + i) / math.sqrt(Cin * k) for i, k in enumerate(ks)]
    bs = [g(dev, Cout, seed=20 + i) for i in range(3)]
    gam, bet = 1 + 0.1 * g(dev, 3 * Cout, seed=30), 0.1 * g(dev, 3 * Cout, seed=31)
    wd, bd = g(dev, Cout, 3 * Cout, seed=40) / 14, g(dev, Cout, seed=41)
    go = g(dev, B, L, Cout, seed=50).to(dev)
    t = lambda a: a.detach().to(dev).requires_grad_()

    def run(only16):
        leaves = [t(x)] + [t(w) for w in ws] + [t(b) for b in bs] + [t(gam), t(bet), t(wd), t(bd)]
        xd, w3, b3, (gd, btd, wdd, bdd) = leaves[0], leaves[1:4], leaves[4:7], leaves[7:]
        y = H.conv_group1d(xd, ks, w3, b3, ln=(gd, btd, 1e-5), out16_only=only16)
        assert H._is16only(y) == only16
        if only16:
            with pytest.raises(RuntimeError):
                H.add(y, y)                      # any other consumer of the placeholder is refused
        z = H.linear(y, wdd, bdd)
        z.backward(go)
        return [z.detach()] + [l.grad for l in leaves]

    a, b = run(True), run(False)
    close(a[0], b[0], tol=0, name="forward")
    for i, (u, v) in enumerate(zip(a[1:], b[1:])):
        close(u, v, tol=1e-2, name=f"grad{i}")


@pytest.mark.parametrize("B,L,Cin,Cout,ks", [(2, 256, 1, 64, (3, 61, 1021)), (3, 512, 64, 128, (3, 31, 251)),
                                             (2, 256, 128, 256, (3, 15, 61)), (2, 64, 256, 64, (3, 11, 31))])
def test_conv_bank_bf16_cat_buffer(dev, bf16_mode, B, L, Cin, Cout, ks):
    """bf16 mode keeps the concatenated conv outputs (the LayerNorm input) in bf16.  Against the fp32
    cat buffer the difference must be one bf16 rounding of the LayerNorm input; against torch (fp32 on
    bf16-rounded x / w) the usual bf16-mode tolerance."""
    from applecider_amd import hipops as H
    rb = lambda t: t.bfloat16().float()
    x = rb(g(dev, B, L, Cin, seed=1))
    ws = [rb(g(dev, Cout, k * Cin, seed=10 + i) / math.sqrt(Cin * k)) for i, k in enumerate(ks)]
    bs = [g(dev, Cout, seed=20 + i) for i in range(3)]
    gam, bet = 1 + 0.1 * g(dev, 3 * Cout, seed=30), 0.1 * g(dev, 3 * Cout, seed=31)
    go = g(dev, B, L, 3 * Cout, seed=50)
    t = lambda a, rg=True: a.detach().to(dev).requires_grad_(rg)

    def run(cat16):
        H._CAT16 = cat16
        try:
            leaves = [t(x, Cin != 1)] + [t(w) for w in ws] + [t(b) for b in bs] + [t(gam), t(bet)]
            y = H.conv_group1d(leaves[0], ks, leaves[1:4], leaves[4:7], ln=(leaves[7], leaves[8], 1e-5))
            y.backward(go.to(dev))
            return [y.detach()] + [l.grad for l in leaves if l.grad is not None]
        finally:
            H._CAT16 = True

    a, b = run(True), run(False)
    for i, (u, v) in enumerate(zip(a, b)):
        close(u, v, tol=2e-2, name=f"cat16 vs fp32 cat buffer, tensor {i}")
    # torch reference of the forward (channels-first convs on the rounded operands)
    xr = x.permute(0, 2, 1)
    ycat = torch.cat([F.conv1d(xr, w.reshape(Cout, k, Cin).permute(0, 2, 1), bb, padding=k // 2)
                      for w, bb, k in zip(ws, bs, ks)], 1).permute(0, 2, 1)
    ref = F.gelu(F.layer_norm(ycat, (3 * Cout,), gam, bet, 1e-5))
    close(a[0], ref, tol=2e-2, name="forward vs torch")


def test_batchnorm_large_mean_many_rows_vs_torch(dev):
    """Round-2 advisor finding: one-pass E[x^2] - mean^2 in fp32 cancels when |mean| >> std at the benchmark's row
    counts.  Columns with means up to 1e3 and std 0.1 .. 1 over 2^20 rows (B*L of a SpectraNet stage), forward
    statistics, output, running statistics and all gradients against F.batch_norm in fp64."""
    import torch.nn.functional as F
    from applecider_amd import hipops as H
    rows, Cn = 1 << 20, 64
    g = torch.Generator().manual_seed(3)
    mean = torch.linspace(-1000.0, 1000.0, Cn)
    std = torch.linspace(0.1, 1.0, Cn)
    x = (torch.randn(rows, Cn, generator=g) * std + mean).to(dev).requires_grad_()
    gamma = (1.0 + 0.1 * torch.randn(Cn, generator=g)).to(dev).requires_grad_()
    beta = (0.1 * torch.randn(Cn, generator=g)).to(dev).requires_grad_()
    rm, rv = torch.zeros(Cn, device=dev), torch.ones(Cn, device=dev)
    go = torch.randn(rows, Cn, generator=g).to(dev)
    y = H.batchnorm_act(x, gamma, beta, rm, rv, True, eps=1e-5, momentum=0.1, act=None)
    y.backward(go)
    x64 = x.detach().double().cpu().requires_grad_()
    g64, b64 = gamma.detach().double().cpu().requires_grad_(), beta.detach().double().cpu().requires_grad_()
    rm64, rv64 = torch.zeros(Cn, dtype=torch.float64), torch.ones(Cn, dtype=torch.float64)
    y64 = F.batch_norm(x64, rm64, rv64, g64, b64, True, 0.1, 1e-5)
    y64.backward(go.double().cpu())
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    # the input itself carries |mean| / std * 2^-24 of rounding (up to 6e-4 of a standard deviation here)
    assert rel(rv, rv64) <= 1e-4, rel(rv, rv64)
    assert rel(rm, rm64) <= 1e-6
    assert rel(y, y64.detach()) <= 1e-3      # x*scale + shift at |x*scale| ~ 1e4: 2^-24 of that per element
    assert rel(x.grad, x64.grad) <= 2e-3
    assert rel(gamma.grad, g64.grad) <= 1e-3
    assert rel(beta.grad, b64.grad) <= 1e-4


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("M,K,N,scale_res", [(512, 3072, 768, True), (4608, 1536, 384, True), (512, 3072, 384, False),
                                             (300, 1024, 100, False),
                                             # nkt = 130 tiles, wanted split 16 -> 9 tiles per piece -> the
                                             # 16th piece would be empty and its slab unwritten (now: 15 pieces)
                                             (128, 4160, 128, False)])  # (M % 4 == 0 for the input-gradient product)
def test_linear_small_grid_split_k(dev, mode, M, K, N, scale_res):
    """Long-reduction products whose output grid leaves most CUs idle (ConvNeXt stage 2 / 3 fc2 and the input
    gradient of fc1, SpectraNet's 3072 -> 384 head; timm block astrominn.py:12-17, spectranet.py:138-155) are cut
    over K and meet in the output with atomics; bias starts the sum, layer scale and skip follow in one pass.
    Same results as the one-workgroup-per-tile form and as torch in fp64, forward and all gradients."""
    from applecider_amd import hipops as H
    H.set_math(mode)
    try:
        gen = torch.Generator().manual_seed(M + N)
        x0 = torch.randn(M, K, generator=gen)
        w0 = torch.randn(N, K, generator=gen) / math.sqrt(K)
        b0 = torch.randn(N, generator=gen)
        cs0 = torch.randn(N, generator=gen) if scale_res else None
        r0 = torch.randn(M, N, generator=gen) if scale_res else None
        go = torch.randn(M, N, generator=gen)
        assert H._small_grid_split(M, N, K) > 1     # (the input gradient splits too when ITS grid is small)
        x64, w64, b64 = (t.double().requires_grad_() for t in (x0, w0, b0))
        y64 = x64 @ w64.t() + b64
        if scale_res:
            cs64, r64 = cs0.double().requires_grad_(), r0.double().requires_grad_()
            y64 = y64 * cs64 + r64
        y64.backward(go.double())
        res = {}
        for split in (True, False):
            H._SMALL_GRID_SPLIT = split
            try:
                x, w, b = (t.clone().to(dev).requires_grad_() for t in (x0, w0, b0))
                cs = cs0.clone().to(dev).requires_grad_() if scale_res else None
                r = r0.clone().to(dev).requires_grad_() if scale_res else None
                y = H.linear(x, w, b, residual=r, colscale=cs)
                y.backward(go.to(dev))
                torch.cuda.synchronize()
            finally:
                H._SMALL_GRID_SPLIT = True
            res[split] = [y.detach(), x.grad, w.grad, b.grad] + ([cs.grad, r.grad] if scale_res else [])
        want = [y64.detach(), x64.grad, w64.grad, b64.grad] + ([cs64.grad, r64.grad] if scale_res else [])
        tol = 2e-5 if mode == "bf16x3" else 5e-6
        for i, (a, b_, w_) in enumerate(zip(res[True], res[False], want)):
            close(a, b_, tol=tol, name=f"split vs plain [{i}]")
            close(a, w_, tol=5e-5, name=f"split vs fp64 [{i}]")
    finally:
        H.set_math("f32")


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_gelu_through_rational_erf_outside_exact_mode(dev, mode):
    """AC_ACT_GELU_FAST (round 4): LayerNorm + GELU, Linear + GELU (epilogue, act' / bias-gradient pass) use the rational
    erf in every mode but the exact one - forward and backward of the same function, within 2e-6 of torch's erf GELU
    beyond the product's own rounding; the exact mode keeps the library erf (<= 5e-7)."""
    from applecider_amd import hipops as H
    H.set_math(mode)
    try:
        assert H._kact(H.ACT_GELU) == (H.ACT_GELU if mode == "f32" else H.ACT_GELU_FAST)
        rows, C = 3000, 384
        x = (g(dev, rows, C, seed=1) * 2 + 0.5).requires_grad_()
        w = (1 + 0.1 * g(dev, C, seed=2)).requires_grad_()
        b = (0.1 * g(dev, C, seed=3)).requires_grad_()
        go = g(dev, rows, C, seed=4)
        x64, w64, b64 = (t.detach().double().requires_grad_() for t in (x, w, b))
        y64 = F.gelu(F.layer_norm(x64, (C,), w64, b64, 1e-6))
        y64.backward(go.double())
        xd, wd, bd = (t.detach().to(dev).requires_grad_() for t in (x, w, b))
        yd = H.layer_norm(xd, wd, bd, 1e-6, act="gelu")
        yd.backward(go.to(dev))
        tol = 1e-6 if mode == "f32" else 3e-6
        close(yd, y64, tol=tol, name="ln+gelu y")
        close(xd.grad, x64.grad, tol=2 * tol, name="ln+gelu dx")
        close(wd.grad, w64.grad, tol=2 * tol, name="ln+gelu dgamma")
        # Linear + GELU: the epilogue and the act' pass
        M, K, N = 2048, 96, 384
        a, wl, bl, gl = g(dev, M, K, seed=5), g(dev, N, K, seed=6) / math.sqrt(K), g(dev, N, seed=7), g(dev, M, N, seed=8)
        a64, wl64, bl64 = (t.double().requires_grad_() for t in (a, wl, bl))
        z64 = F.gelu(a64 @ wl64.t() + bl64)
        z64.backward(gl.double())
        ad, wld, bld = (t.clone().to(dev).requires_grad_() for t in (a, wl, bl))
        zd = H.linear(ad, wld, bld, act="gelu")
        zd.backward(gl.to(dev))
        ptol = 2e-6 if mode == "f32" else 3e-5       # the product itself in split-bf16 mode
        close(zd, z64, tol=ptol, name="linear+gelu")
        close(ad.grad, a64.grad, tol=2 * ptol, name="linear+gelu dx")
        close(wld.grad, wl64.grad, tol=2 * ptol, name="linear+gelu dw")
        close(bld.grad, bl64.grad, tol=2 * ptol, name="linear+gelu db")
    finally:
        H.set_math("f32")


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("B,H_,C", [(513, 15, 96), (343, 7, 192), (64, 3, 384), (33, 1, 768), (5, 15, 96), (4, 7, 40), (3, 9, 32)])
def test_dwconv_with_the_block_shortcut(dev, B, H_, C, variant):
    """ConvNeXt block (timm block at astrominn.py:12-17): out = x + f(dw(x)).  dwconv7x7_shortcut returns (dw(x), x) and
    its backward kernel adds the shortcut's gradient where it stores dx (ac_dwconv7x7_bwd_res; every kernel family:
    pipelined 15x15 / 7x7, small 3x3 / 1x1, whole-row, generic) - equal to torch's autograd sum of the two paths."""
    from applecider_amd import hipops as H
    x = g(dev, B, C, H_, H_, seed=1).requires_grad_()
    w = (g(dev, C, 1, 7, 7, seed=2) / 7).requires_grad_()
    b = g(dev, C, seed=3).requires_grad_()
    y = F.conv2d(x, w, b, padding=3, groups=C)
    out = x * 0.5 + torch.tanh(y)                      # a shortcut with its own scale and a nonlinear branch
    go = g(dev, *y.shape, seed=4)
    out.backward(go)
    H._DWCONV_VARIANT = variant
    try:
        xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_()
        wd = w.detach().reshape(C, 49).t().contiguous().to(dev).requires_grad_()
        bd = b.detach().to(dev).requires_grad_()
        yd, sc = H.dwconv7x7_shortcut(xd, wd, bd)
        od = sc * 0.5 + torch.tanh(yd)
        od.backward(go.permute(0, 2, 3, 1).contiguous().to(dev))
        torch.cuda.synchronize()
    finally:
        H._DWCONV_VARIANT = 0
    close(od.permute(0, 3, 1, 2), out, name="out")
    close(xd.grad.permute(0, 3, 1, 2), x.grad, name="dx (both paths)")
    close(wd.grad.t().reshape(C, 1, 7, 7), w.grad, name="dw")
    close(bd.grad, b.grad, name="db")


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("mname,M,N,K,split", [("TN", 384, 1536, 4608, 8), ("TN", 96, 384, 20000, 40), ("NT", 1000, 96, 384, 1)])
def test_gemm_grouped_equals_separate_launches(dev, mode, mname, M, N, K, split):
    """ac_gemm_grouped: up to 16 same-shaped products in one launch, operands from a pointer table passed by value;
    split-K pieces meet with atomics.  Measured in the step and NOT used by it (deferring the ConvNeXt / encoder weight
    gradients into grouped launches cost 0.55 ms of overlap, profiles/r04_ab_switches.txt): the entry point stays for
    hosts that have independent same-shaped products at hand."""
    from applecider_amd import hipops as H
    H.set_math(mode)
    try:
        torch.manual_seed(M + N)
        G = 5
        md = {"TN": H.AC_GEMM_TN, "NT": H.AC_GEMM_NT}[mname]
        if mname == "TN":
            As = [torch.randn(K, M, device=dev) for _ in range(G)]
            Bs = [torch.randn(K, N, device=dev) for _ in range(G)]
            la, lb = M, N
        else:
            As = [torch.randn(M, K, device=dev) for _ in range(G)]
            Bs = [torch.randn(N, K, device=dev) for _ in range(G)]
            la, lb = K, K
        acc = 2 if split > 1 else 0
        c1 = [torch.zeros(M, N, device=dev) for _ in range(G)]
        c2 = [torch.zeros(M, N, device=dev) for _ in range(G)]
        for a, b, c in zip(As, Bs, c1):
            H.gemm(md, M, N, K, H.mat(H._p(a), la), H.mat(H._p(b), lb), H.mat(H._p(c), N), accumulate=acc, split_k=split)
        H.gemm(md, M, N, K, H.mat(H._p(As[0]), la), H.mat(H._p(Bs[0]), lb), H.mat(H._p(c2[0]), N), accumulate=acc,
               split_k=split, group=[(H._p(a), H._p(b), H._p(c)) for a, b, c in zip(As, Bs, c2)])
        torch.cuda.synchronize()
        for i in range(G):
            assert float(c1[i].abs().max()) > 0
            if split > 1:
                close(c2[i], c1[i], tol=2e-6, name=f"product {i}")
            else:
                assert torch.equal(c2[i], c1[i]), f"product {i}"
    finally:
        H.set_math("f32")


@pytest.mark.parametrize("M,N,K", [(4608, 1536, 384), (1000, 96, 384), (66048, 512, 128), (260, 40, 72), (512, 384, 3072)])
def test_linear_with_plane_fed_weight_equals_on_the_fly_split(dev, M, N, K):
    """Split-bf16 nn.Linear products read their weight as cached (hi, lo) planes (ac_gemm_desc.b_hi / b_lo: split once
    per optimizer step instead of in every workgroup's K loop).  The planes hold exactly what the on-the-fly split
    computes, so forward and input gradient are BIT-IDENTICAL to the fp32-operand form - ragged M / N, the split-K
    forms (slabs forward, atomics backward) included; a parameter living in a flat buffer takes its planes from the
    buffer's one-launch split."""
    from applecider_amd import hipops as H
    from applecider_amd.optim import FlatAdam
    H.set_math("bf16x3")
    try:
        torch.manual_seed(M + N)
        x0 = torch.randn(M, K, device=dev)
        go = torch.randn(M, N, device=dev)
        res = {}
        for planes in (True, False, "flat"):
            w = torch.nn.Parameter(torch.randn(N, K, generator=torch.Generator().manual_seed(5)).to(dev) / math.sqrt(K))
            b = torch.nn.Parameter(torch.randn(N, generator=torch.Generator().manual_seed(6)).to(dev))
            if planes == "flat":
                opt = FlatAdam([{"params": [w, b]}], lr=1e-3)
                opt.prepare()
                assert H._plane_mirror(w) is not None
            H._PLANE_B = bool(planes)
            H.clear_step_cache()
            try:
                x = x0.clone().requires_grad_()
                y = H.linear(x, w, b, act="gelu" if N % 2 == 0 and M != 512 else None)
                y.backward(go)
                torch.cuda.synchronize()
            finally:
                H._PLANE_B = True
            res[planes] = (y.detach().clone(), x.grad.clone())
        split_bwd = H._small_grid_split(M, K, N) > 1           # atomics: equal to rounding, not bit for bit
        for tag in (True, "flat"):
            assert torch.equal(res[tag][0], res[False][0]), f"forward differs ({tag})"
            if split_bwd:
                close(res[tag][1], res[False][1], tol=2e-6, name="dx")
            else:
                assert torch.equal(res[tag][1], res[False][1]), f"dx differs ({tag})"
    finally:
        H.set_math("f32")


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("act,p1,M,K,Hd,N", [("gelu", 0.0, 2304, 96, 384, 96), ("relu", 0.3, 1032, 128, 512, 128),
                                             ("relu", 0.0, 700, 128, 512, 128), ("gelu", 0.0, 260, 40, 72, 24)])
def test_activation_backward_in_the_consumers_product(dev, mode, act, p1, M, K, Hd, N):
    """Round 4: y = fc2(drop(act(fc1(x)))) - fc2's input-gradient product applies act' (+ the dropout mask of fc1's
    epilogue, same seed and index) in its epilogue and sums fc1's bias gradient into its sink (ac_gemm_desc.dact /
    colsum), instead of a separate pass over the hidden tensor.  Every gradient equals the unfused path's (bias sums by
    atomics: to rounding) and fp64 autograd."""
    import itertools
    from applecider_amd import hipops as H
    H.set_math(mode)
    try:
        torch.manual_seed(11)
        x0 = torch.randn(M, K, device=dev)
        go = torch.randn(M, N, device=dev)
        ps0 = [torch.randn(Hd, K, device=dev) / math.sqrt(K), torch.randn(Hd, device=dev) * 0.1,
               torch.randn(N, Hd, device=dev) / math.sqrt(Hd), torch.randn(N, device=dev) * 0.1]
        res = {}
        for fused in (True, False):
            H._FUSE_DACT = fused
            H._seed_counter = itertools.count(4000)
            try:
                ps = [torch.nn.Parameter(t.clone()) for t in ps0]
                for prm in ps:
                    prm.grad = torch.zeros_like(prm)
                x = x0.clone().requires_grad_()
                y = H.mlp(x, ps[0], ps[1], ps[2], ps[3], act, p1=p1, p2=0.0, training=True, residual=None)
                y.backward(go)
                torch.cuda.synchronize()
            finally:
                H._FUSE_DACT = True
            res[fused] = [y.detach().clone(), x.grad.clone()] + [prm.grad.clone() for prm in ps]
        tol = 2e-6 if mode == "f32" else 4e-6
        assert torch.equal(res[True][0], res[False][0])
        for i, (a, b) in enumerate(zip(res[True][1:], res[False][1:])):
            close(a, b, tol=tol, name=f"fused vs separate [{i}]")
        if p1 == 0.0 and act == "gelu":      # (ReLU: a gate within rounding of zero flips between fp64 and the GPU product)
            x64 = x0.double().cpu().requires_grad_()
            w64 = [t.double().cpu().requires_grad_() for t in ps0]
            h = x64 @ w64[0].t() + w64[1]
            h = F.gelu(h) if act == "gelu" else F.relu(h)
            y64 = h @ w64[2].t() + w64[3]
            y64.backward(go.double().cpu())
            ptol = 5e-6 if mode == "f32" else 5e-5
            for i, (a, b) in enumerate(zip(res[True][1:], [x64.grad] + [t.grad for t in w64])):
                close(a, b, tol=ptol, name=f"fused vs fp64 [{i}]")
    finally:
        H.set_math("f32")


def test_fused_activation_backward_refuses_a_shared_hidden_tensor(dev):
    """The fusion assumes the activated output has ONE consumer; a second consumer makes autograd sum two gradients, one
    of which never saw act' - that must fail loudly, not silently."""
    from applecider_amd import hipops as H
    torch.manual_seed(1)
    x = torch.randn(512, 64, device=dev, requires_grad=True)
    w1, w2 = torch.randn(128, 64, device=dev, requires_grad=True), torch.randn(64, 128, device=dev, requires_grad=True)
    h = H.linear(x, w1, None, act="gelu")
    y = H.linear(h, w2, None)
    with pytest.raises(RuntimeError, match="fused activation backward"):
        (y.sum() + h.sum()).backward()


def test_zero_pool_hands_out_disjoint_zeroed_slices(dev):
    """Backward's zeroed temporaries are slices of one zero-filled buffer per stream and step (hipops._zeros): every slice
    is zero when handed out, no two overlap, a slice outlives its pool buffer, a new step starts a fresh buffer sized to
    the last step's demand, and nothing is pooled while a graph is being captured."""
    from applecider_amd import hipops as H
    H.zero_pools_new_step()
    a = H._zeros((1000, 3), dev)
    b = H._zeros((7,), dev)
    c = H._zeros((5 << 20,), dev)                       # larger than what is left: a fresh buffer
    assert a.shape == (1000, 3) and a.dtype == torch.float32 and a.is_contiguous()
    for t in (a, b, c):
        assert t.data_ptr() % 256 == 0 and not t.any()
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * 4) for t in (a, b, c))
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(2))
    a.fill_(1.0)
    b.fill_(2.0)
    H.zero_pools_new_step()
    d = H._zeros((1000, 3), dev)
    assert not d.any() and float(a.sum()) == 3000.0 and float(b.sum()) == 14.0     # old slices untouched, still alive
    pool = H._zero_pools[(dev.index if dev.index is not None else torch.cuda.current_device(), H._stream())]
    assert pool.target >= (5 << 20)                      # sized to the last step's demand
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        e = H._zeros((64,), dev)
        assert not e.any()
    assert len(H._zero_pools) >= 2                       # one pool per stream
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f = H._zeros((64,), dev)                         # plain torch.zeros inside a capture
    g.replay()
    torch.cuda.synchronize()
    assert not f.any()
