"""LDS-window weight-gradient kernel (csrc/ac_wgrad.hip) against torch's conv1d weight gradient
(spectranet.py:18-20: 'same' Conv1d, padding k//2) in fp64, for bf16 operands and split-bf16 planes."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(B, L, Cin, Cout, k, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, Cin, generator=g)
    dy = torch.randn(B, L, Cout, generator=g)
    return x, dy


def _ref(x, dy, k):
    # dW[co, ci, t] of y = conv1d(x^T, w, padding=k//2), upstream gradient dy^T
    xt = x.double().permute(0, 2, 1)
    w = torch.zeros(dy.shape[2], x.shape[2], k, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(xt, w, padding=k // 2)
    (gw,) = torch.autograd.grad(y, w, dy.double().permute(0, 2, 1))
    return gw.permute(0, 2, 1).reshape(dy.shape[2], k * x.shape[2])   # tap-major [Cout, k*Cin]


@pytest.mark.parametrize("B,L,Cin,Cout,k", [(2, 128, 64, 128, 31), (3, 64, 128, 256, 7), (2, 256, 64, 128, 251),
                                            (1, 64, 64, 128, 13),
                                            # short sequences: 64 / L whole samples per 64-position K step
                                            (8, 16, 128, 128, 13), (12, 16, 64, 256, 7), (6, 32, 64, 128, 11),
                                            (4, 16, 64, 128, 9)])
@pytest.mark.parametrize("split", [False, True])
def test_conv_wgrad_window(dev, B, L, Cin, Cout, k, split):
    from applecider_amd import hipops as H
    if not split and k < 7:
        pytest.skip("bf16 operands: short kernels stay on the generic TN product")
    x, dy = _case(B, L, Cin, Cout, k, seed=k)
    P = k // 2 + 3                       # extra padding rows: row_base != 0 on both operands
    Lp = L + 2 * P
    xd, dyd = x.to(dev), dy.to(dev)
    if split:
        xh, xl = H._pad_rows_split(xd, B, L, Cin, P, Lp)
        dyh, dyl = H._pad_rows_split(dyd, B, L, Cout, 2, L + 4)
        want = _ref(x, dy, k)
        tol = 3e-5
    else:
        xh, xl = H._pad_rows16(xd, B, L, Cin, P, Lp), None
        dyh, dyl = H._pad_rows16(dyd, B, L, Cout, 2, L + 4), None
        want = _ref(x.bfloat16().float(), dy.bfloat16().float(), k)     # exact up to summation order
        tol = 2e-4
    dw = torch.zeros(Cout, k * Cin, device=dev)
    ok = H.conv_wgrad(dyh, dyl, (L + 4) * Cout, Cout, 2, 0, xh, xl, Lp * Cin, Cin, P - k // 2, Lp, B, L, Cout,
                      Cin, k, dw)
    assert ok
    err = float((dw.cpu().double() - want).abs().max() / want.abs().max())
    assert err <= tol, err
    # accumulation contract: a second call adds to dw
    assert H.conv_wgrad(dyh, dyl, (L + 4) * Cout, Cout, 2, 0, xh, xl, Lp * Cin, Cin, P - k // 2, Lp, B, L, Cout,
                        Cin, k, dw)
    err2 = float((dw.cpu().double() - 2 * want).abs().max() / want.abs().max())
    assert err2 <= 2 * tol, err2


def test_conv_wgrad_refuses_uncovered_shapes(dev):
    from applecider_amd import hipops as H
    z = torch.zeros(8, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(8, device=dev)
    assert not H.conv_wgrad(z, None, 0, 0, 0, 0, z, None, 0, 0, 0, 1, 1, 16, 128, 64, 31, dw)    # B * L % 64
    assert not H.conv_wgrad(z, None, 0, 0, 0, 0, z, None, 0, 0, 0, 1, 8, 8, 128, 64, 31, dw)     # L = 8
    assert not H.conv_wgrad(z, None, 0, 0, 0, 0, z, None, 0, 0, 0, 1, 1, 64, 128, 64, 3, dw)     # k < 7
    assert not H.conv_wgrad(z, None, 0, 0, 0, 0, z, None, 0, 0, 0, 1, 1, 64, 96, 64, 31, dw)     # Cout % 128
