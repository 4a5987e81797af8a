"""Matrix-core attention (csrc/ac_attn.hip) against (a) a torch fp64 restatement of the masked
softmax attention inside nn.TransformerEncoderLayer (HyraxBaselineCLS.py:24-31,73-79) and (b) the
scalar fp32 kernels of ac_seq.hip under the SAME dropout seed (both kernel families share the counter
RNG and index, so masks are identical)."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv, pad, H_, go):
    B, T, D3 = qkv.shape
    D, Dh = D3 // 3, D3 // 3 // H_
    x = qkv.double().requires_grad_()
    q, k, v = x.split(D, -1)
    sh = lambda t: t.reshape(B, T, H_, Dh).permute(0, 2, 1, 3)
    s = (sh(q) / math.sqrt(Dh)) @ sh(k).transpose(-1, -2)
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, -1) @ sh(v)).permute(0, 2, 1, 3).reshape(B, T, D)
    o.backward(go.double())
    return o.detach(), x.grad


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


@pytest.fixture
def mode(request):
    from applecider_amd import hipops as H
    H.set_math(request.param)
    yield request.param
    H.set_math("f32")


@pytest.mark.parametrize("mode,tol", [("bf16x3", 5e-5), ("bf16", 2e-2)], indirect=["mode"])
@pytest.mark.parametrize("B,T,H_", [(3, 129, 8), (2, 258, 8), (4, 20, 4), (2, 33, 8), (1, 288, 2)])
def test_mha_mfma_vs_fp64(dev, mode, tol, B, T, H_):
    from applecider_amd import hipops as H
    D = H_ * 16
    gen = torch.Generator().manual_seed(T)
    qkv = torch.randn(B, T, 3 * D, generator=gen)
    go = torch.randn(B, T, D, generator=gen)
    lens = [T, max(2, T // 3), 1, 7][:B]
    pad = torch.zeros(B, T, dtype=torch.bool)
    for i, n in enumerate(lens):
        pad[i, n:] = True
    o, dqkv = _ref(qkv, pad, H_, go)
    qd = qkv.to(dev).requires_grad_()
    od = H.mha(qd, pad.to(torch.uint8).to(dev), H_)
    od.backward(go.to(dev))
    assert _rel(od, o) <= tol, ("out", _rel(od, o))
    assert _rel(qd.grad, dqkv) <= 2 * tol, ("dqkv", _rel(qd.grad, dqkv))
    # gradients of fully padded keys are exactly zero (they never enter a softmax)
    for i, n in enumerate(lens):
        assert qd.grad[i, n:, D:].abs().max().item() == 0.0 if n < T else True


@pytest.mark.parametrize("mode", ["bf16x3"], indirect=True)
def test_mha_mfma_dropout_matches_scalar_kernels(dev, mode):
    """Training path (dropout 0.4 on the attention weights): the matrix-core kernels and the scalar
    fp32 kernels draw the same mask from the same seed, forward and backward."""
    from applecider_amd import hipops as H
    B, T, H_, D = 3, 129, 8, 128
    gen = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, T, 3 * D, generator=gen).to(dev)
    go = torch.randn(B, T, D, generator=gen).to(dev)
    pad = torch.zeros(B, T, dtype=torch.uint8)
    pad[1, 40:] = 1
    pad[2, 1:] = 1
    pad = pad.to(dev)
    seed = 987654321
    res = {}
    for name, flag in (("mfma", True), ("scalar", False)):
        H._MHA_MFMA = flag
        try:
            q = qkv.clone().requires_grad_()
            o = H._MHA.apply(q, pad, H_, 0.4, seed)
            o.backward(go)
            res[name] = (o.detach(), q.grad)
        finally:
            H._MHA_MFMA = True
    assert _rel(res["mfma"][0], res["scalar"][0]) <= 1e-4
    assert _rel(res["mfma"][1], res["scalar"][1]) <= 2e-4
    # and dropout really is on: the output differs from the p = 0 output
    o0 = H._MHA.apply(qkv, pad, H_, 0.0, 0)
    assert _rel(res["mfma"][0], o0) > 0.05
