"""Grouped fused ResidualTowerBlock kernels (ac_tower_blocks_fwd / _bwd; astrominn.py:44-64, 94-131, 249-295).

* single blocks through the fused launch against the reference's golden g1 (output, input gradient, weight
  gradients) — the same fixture the per-op module path is held to;
* fused launch == per-op module path (eval mode) for the three block geometries, a ragged batch (B not a
  multiple of the 16-sample slice), with and without the metadata column gather, every parameter gradient;
* dropout on: the backward of the fused launch is the derivative of its forward under the same seed
  (directional finite difference), the output differs from eval, and two seeds differ;
* AstroMiNN end to end: fused towers + experts against the per-op path, logits and all gradients.
"""
import itertools

import numpy as np
import pytest
import torch

from common import T, assert_close, cfg_default, closed_form_sd, gold

pytestmark = pytest.mark.gpu

LOGIT_TOL, GRAD_TOL = 1e-3, 2e-3


@pytest.fixture()
def dev():
    from applecider_amd import hipops as H
    H.set_math("f32")
    H.FUSED_TOWERS = True
    yield torch.device("cuda:0")
    H.FUSED_TOWERS = True
    H.set_math("f32")


def build(cls, cfg, dev, salt=0):
    m = cls(cfg)
    m.load_state_dict(closed_form_sd(m, salt))
    return m.to(dev)


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _plan_for(H, blocks, cols=None, need_dx=True, width=None):
    from applecider_amd.models.astrominn import AstroMiNN
    bl, off = [], 0
    for b in blocks:
        n_out = b.main_path[2].out_features
        bl.append({"n_in": b.start_path[0].in_features, "hid": b.start_path[0].out_features, "n_out": n_out,
                   "eps": b.main_path[0].eps, "cols": cols, "y_off": off, "ldy": None})
        off += n_out
    for d in bl:
        d["ldy"] = off
    params = [t for b in blocks for t in AstroMiNN._block_params(b)]
    return H.TowerPlan(bl, lambda B: (B, off), need_dx=need_dx), params


@pytest.mark.parametrize("tag,dims", [("a", (2, 16, 32)), ("b", (19, 128, 32)), ("c", (288, 128, 5))])
def test_fused_block_vs_reference_golden(dev, tag, dims):
    from applecider_amd import hipops as H
    from applecider_amd.models.astrominn import ResidualTowerBlock
    g = gold("g1_towers.npz")
    i, h, o = dims
    m = ResidualTowerBlock(i, h, o)
    m.load_state_dict(closed_form_sd(m))
    m = m.to(dev).eval()
    x = T(np.random.default_rng(100 + i).standard_normal((8, i)).astype(np.float32)).to(dev).requires_grad_()
    plan, params = _plan_for(H, [m])
    y = H.tower_blocks(x, None, plan, False, params)
    assert_close(y, g[f"{tag}.y"], LOGIT_TOL, "y")
    (dx,) = torch.autograd.grad(y, x, 2.0 * y.detach(), retain_graph=True)
    assert_close(dx, g[f"{tag}.dx"], GRAD_TOL, "dx")
    gw = torch.autograd.grad(y, [m.start_path[0].weight, m.activation[2].weight], 2.0 * y.detach())
    assert_close(gw[0], g[f"{tag}.dw_start"], GRAD_TOL, "dw start_path.0")
    assert_close(gw[1], g[f"{tag}.dw_act"], GRAD_TOL, "dw activation.2")


@pytest.mark.parametrize("B", [37, 64])
def test_fused_group_equals_per_op_modules(dev, B):
    """Three blocks of different geometry in ONE launch (one of them with an identity skip) against the
    per-op module path, every parameter gradient and dL/dx."""
    from applecider_amd import hipops as H
    from applecider_amd.models.astrominn import ResidualTowerBlock
    torch.manual_seed(3)
    blocks = [ResidualTowerBlock(24, 16, 32).to(dev).eval(), ResidualTowerBlock(24, 128, 32).to(dev).eval(),
              ResidualTowerBlock(24, 48, 24).to(dev).eval()]      # 24 -> 24: identity skip
    for b in blocks:   # LayerNorm affines away from (1, 0) so their gradients are exercised
        for ln in (b.main_path[0], b.activation[0]):
            ln.weight.data.uniform_(0.5, 1.5)
            ln.bias.data.uniform_(-0.3, 0.3)
    x = torch.randn(B, 24, device=dev, requires_grad=True)
    r = torch.randn(B, 88, device=dev)
    ref = torch.cat([b(x) for b in blocks], dim=1)
    plist = [p for b in blocks for p in b.parameters()]
    gref = torch.autograd.grad((ref * r).sum(), [x] + plist)
    plan, params = _plan_for(H, blocks)
    y = H.tower_blocks(x, None, plan, False, params)
    assert _rel(y, ref) <= 2e-6
    gfu = torch.autograd.grad((y * r).sum(), [x] + plist)
    for a, b_, p in zip(gfu, gref, ["x"] + [n for b in blocks for n, _ in b.named_parameters()]):
        assert _rel(a, b_) <= 2e-5, p


def test_fused_group_with_column_gather_and_extra_slot(dev):
    from applecider_amd import hipops as H
    from applecider_amd.models.astrominn import AstroMiNN, ResidualTowerBlock
    torch.manual_seed(4)
    cols = [6, 9, 10, 13, 15, 17, 18]
    blk = ResidualTowerBlock(7, 32, 32).to(dev).eval()
    md = torch.randn(21, 24, device=dev)
    extra = torch.randn(21, 32, device=dev, requires_grad=True)
    plan = H.TowerPlan([{"n_in": 7, "hid": 32, "n_out": 32, "eps": 1e-5, "cols": cols, "y_off": 32, "ldy": 64}],
                       lambda B: (B, 64), extra_off=0, need_dx=False)
    y = H.tower_blocks(md, extra, plan, False, AstroMiNN._block_params(blk))
    ref = torch.cat([extra, blk(md[:, cols].contiguous())], dim=1)
    assert _rel(y, ref) <= 2e-6
    r = torch.randn_like(y)
    (de,) = torch.autograd.grad((y * r).sum(), extra)
    assert torch.equal(de, r[:, :32])


def test_fused_dropout_backward_is_the_derivative_of_its_forward(dev):
    from applecider_amd import hipops as H
    from applecider_amd.models.astrominn import ResidualTowerBlock
    torch.manual_seed(5)
    blocks = [ResidualTowerBlock(288, 128, 5).to(dev).train() for _ in range(2)]
    plan, params = _plan_for(H, blocks)
    x = torch.randn(40, 288, device=dev)
    v = torch.randn_like(x)
    r = torch.randn(40, 10, device=dev)

    def f(xx, seed0=700):
        H._seed_counter = itertools.count(seed0)      # the same host seed for every evaluation
        return H.tower_blocks(xx, None, plan, True, params)

    xg = x.clone().requires_grad_()
    y = f(xg)
    (dx,) = torch.autograd.grad((y * r).sum(), xg)
    eps = 1e-2
    fd = ((f(x + eps * v) * r).sum() - (f(x - eps * v) * r).sum()) / (2 * eps)
    an = (dx * v).sum()
    assert abs(float(fd) - float(an)) <= 5e-3 * max(1.0, abs(float(an))), (float(fd), float(an))
    y_eval = H.tower_blocks(x, None, plan, False, params)
    assert _rel(y.detach(), y_eval) > 1e-2                 # dropout really is on
    assert torch.equal(f(x), y.detach())                   # deterministic under one seed
    assert not torch.equal(f(x, 701), y.detach())          # another seed, another mask


def test_astrominn_fused_towers_equal_per_op_path(dev):
    from applecider_amd import hipops as H
    from applecider_amd.models.astrominn import AstroMiNN
    from applecider_amd.synthetic import make_batch
    m = build(AstroMiNN, cfg_default(), dev).eval()
    b = make_batch(24, seed=1)
    batch = tuple(T(b[k]).to(dev) for k in ("metadata", "image", "target"))
    out = {}
    for fused in (True, False):
        H.FUSED_TOWERS = fused
        m.this_optimizer.zero_grad()
        logits = m(batch)
        loss = m.this_criterion(logits, batch[2])
        loss.backward()
        out[fused] = (logits.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
    assert m._plans() is not None
    assert _rel(out[True][0], out[False][0]) <= 2e-6
    assert torch.equal(out[True][0].argmax(1), out[False][0].argmax(1))
    for n, g in out[False][1].items():
        if g.abs().max() > 0:
            assert _rel(out[True][1][n], g) <= 5e-5, n
