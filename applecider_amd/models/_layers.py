"""Leaf modules of the MI355X path.  Parameter NAMES and state_dict SHAPES are those of the torch /
timm modules the reference uses, so checkpoints interchange; the forward of every leaf is a HIP
kernel call through applecider_amd.hipops (no ATen compute, no CPU fallback).

Leaves that keep their weight in a kernel-friendly layout (tap-major conv weights, zero-padded
input projections) convert on state_dict save/load, so the checkpoint layout stays the
reference's.
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import hipops as H


def _init_linear_(weight, bias, fan_in):
    # nn.Linear / nn.ConvNd default: kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    with torch.no_grad():
        weight.uniform_(-bound, bound)
        if bias is not None:
            bias.uniform_(-bound, bound)


class Linear(nn.Module):
    """nn.Linear with a fused epilogue (activation / layer-scale / residual)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        _init_linear_(self.weight, self.bias, in_features)

    def forward(self, x, act=None, residual=None, colscale=None):
        return H.linear(x, self.weight, self.bias, act=act, residual=residual, colscale=colscale)


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x, act=None):
        return H.layer_norm(x, self.weight, self.bias, self.eps, act=act)


class BatchNorm1d(nn.Module):
    """nn.BatchNorm1d(C) for channels-last sequences [B, L, C] (statistics over B*L per channel), fused
    with the activation that follows it; same parameters / buffers / state_dict keys as torch's module."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, act=None):
        if self.training:
            self.num_batches_tracked += 1
        return H.batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                               self.eps, self.momentum, act)


class Dropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        return H.dropout(x, self.p, self.training)


class Marker(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential indices equal to the reference's
    (GELU / ReLU / Tanh / Sigmoid slots); the activation itself is fused into a neighbour kernel."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind

    def forward(self, x):
        return H.activation(x, self.kind)

    def extra_repr(self):
        return self.kind


class _LayoutLeaf(nn.Module):
    """Base for leaves whose `weight` is stored in kernel layout.  Subclasses implement
    to_kernel(ref_weight) and to_reference(kernel_weight)."""

    def to_kernel(self, w):
        raise NotImplementedError

    def to_reference(self, w):
        raise NotImplementedError

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        key = prefix + "weight"
        if key in destination:
            w = destination[key]
            destination[key] = self.to_reference(w if keep_vars else w.detach())

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        key = prefix + "weight"
        if key in state_dict and tuple(state_dict[key].shape) != tuple(self.weight.shape):
            state_dict = dict(state_dict)
            state_dict[key] = self.to_kernel(state_dict[key])
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)


class StemConv4x4(_LayoutLeaf):
    """Conv2d(in_chans, 96, 4, stride 4) as patchify + GEMM.  Checkpoint weight [Cout,Cin,4,4];
    kernel weight [Cout, 64] in (ky, kx, c) order, zero padded from 16*Cin."""

    def __init__(self, in_chans, out_chans):
        super().__init__()
        assert in_chans <= 4
        self.in_chans, self.out_chans = in_chans, out_chans
        self.weight = nn.Parameter(torch.zeros(out_chans, 64))
        self.bias = nn.Parameter(torch.zeros(out_chans))
        w = torch.empty(out_chans, in_chans, 4, 4)
        _init_linear_(w, self.bias, in_chans * 16)
        with torch.no_grad():
            self.weight.copy_(self.to_kernel(w))

    def to_kernel(self, w):
        out = w.new_zeros(self.out_chans, 64)
        out[:, :16 * self.in_chans] = w.permute(0, 2, 3, 1).reshape(self.out_chans, -1)
        return out

    def to_reference(self, w):
        return w[:, :16 * self.in_chans].reshape(self.out_chans, 4, 4, self.in_chans).permute(0, 3, 1, 2).contiguous()

    def forward(self, img):
        patches, OH, OW = H.stem_patchify(img)
        y = H.linear(patches, self.weight, self.bias)
        return y.reshape(img.shape[0], OH, OW, self.out_chans)


class DWConv7x7(_LayoutLeaf):
    """Depthwise Conv2d(C, C, 7, padding 3, groups C) on NHWC.  Checkpoint [C,1,7,7]; kernel [49,C]."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.weight = nn.Parameter(torch.empty(49, dim))
        self.bias = nn.Parameter(torch.empty(dim))
        _init_linear_(self.weight, self.bias, 49)

    def to_kernel(self, w):
        return w.reshape(self.dim, 49).t().contiguous()

    def to_reference(self, w):
        return w.t().reshape(self.dim, 1, 7, 7).contiguous()

    def forward(self, x):
        return H.dwconv7x7(x, self.weight, self.bias)


class PatchConv2x2(_LayoutLeaf):
    """Conv2d(Cin, Cout, 2, stride 2) on NHWC.  Checkpoint [Cout,Cin,2,2]; kernel [Cout,(ky,kx,ci)]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, 4 * cin))
        self.bias = nn.Parameter(torch.empty(cout))
        _init_linear_(self.weight, self.bias, 4 * cin)

    def to_kernel(self, w):
        return w.permute(0, 2, 3, 1).reshape(self.cout, 4 * self.cin).contiguous()

    def to_reference(self, w):
        return w.reshape(self.cout, 2, 2, self.cin).permute(0, 3, 1, 2).contiguous()

    def forward(self, x):
        return H.patch_conv2x2(x, self.weight, self.bias)


class Conv1dTap(_LayoutLeaf):
    """nn.Conv1d(Cin, Cout, k, padding=k//2) parameters.  Checkpoint [Cout,Cin,k]; kernel layout
    [Cout, k*Cin] tap-major (the im2col row of the implicit GEMM).  Run through conv_group1d."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.weight = nn.Parameter(torch.empty(cout, k * cin))
        self.bias = nn.Parameter(torch.empty(cout))
        _init_linear_(self.weight, self.bias, cin * k)

    def to_kernel(self, w):
        return w.permute(0, 2, 1).reshape(self.cout, self.k * self.cin).contiguous()

    def to_reference(self, w):
        return w.reshape(self.cout, self.k, self.cin).permute(0, 2, 1).contiguous()


class PointConv1d(_LayoutLeaf):
    """nn.Conv1d(Cin, Cout, 1) on [B, L, C] = a Linear.  Checkpoint [Cout,Cin,1]; kernel [Cout,Cin]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout))
        _init_linear_(self.weight, self.bias, cin)

    def to_kernel(self, w):
        return w.reshape(self.cout, self.cin).contiguous()

    def to_reference(self, w):
        return w.reshape(self.cout, self.cin, 1)

    def forward(self, x):
        return H.linear(x, self.weight, self.bias)


class InProj8(_LayoutLeaf):
    """nn.Linear(7, d_model) with the input padded to 8 channels (16-byte rows).
    Checkpoint [D,7]; kernel [D,8] with a zero last column."""

    def __init__(self, in_features, out_features):
        super().__init__()
        assert in_features <= 8
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.zeros(out_features, 8))
        self.bias = nn.Parameter(torch.empty(out_features))
        w = torch.empty(out_features, in_features)
        _init_linear_(w, self.bias, in_features)
        with torch.no_grad():
            self.weight[:, :in_features] = w

    def to_kernel(self, w):
        out = w.new_zeros(self.out_features, 8)
        out[:, :self.in_features] = w
        return out

    def to_reference(self, w):
        return w[:, :self.in_features].contiguous()
