"""ConvNeXt-Tiny backbone on the MI355X path (replaces timm.create_model("convnext_tiny",
pretrained=False, in_chans=3, num_classes=0) called at src/applecider/models/astrominn.py:12-17).

timm is a third-party dependency that is absent here, so the architecture is restated from its
published definition (stem Conv4x4/s4 + LN; depths [3,3,9,3], dims [96,192,384,768]; block =
dw7x7 -> LN(1e-6) -> Linear(4C) -> GELU(erf) -> Linear(C) -> * gamma -> + shortcut; LN + Conv2x2/s2
between stages; avg-pool -> head.norm -> flatten) with timm 1.0.x state_dict names.

Activations are NHWC end to end ([B, H, W, C] = rows of C for every pointwise op):
  stem       ac_stem_patchify + gather-GEMM          (K = 48 padded to 64)
  dw7x7      ac_dwconv7x7 (LDS-resident 32-channel planes)
  LN         ac_layernorm (row kernel)
  fc1+GELU   ac_gemm epilogue (pre-activation saved for backward)
  fc2        ac_gemm epilogue: * gamma + shortcut
  downsample ac_layernorm + 2x2 patch gather-GEMM
"""

from __future__ import annotations

import torch
import torch.nn as nn

from .. import hipops as H
from ._layers import DWConv7x7, LayerNorm, Linear, PatchConv2x2, StemConv4x4

DEPTHS = (3, 3, 9, 3)
DIMS = (96, 192, 384, 768)


def _trunc_normal_(t, std=0.02):
    with torch.no_grad():
        nn.init.trunc_normal_(t, std=std)


class Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = Linear(dim, 4 * dim)
        self.fc2 = Linear(4 * dim, dim)


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))
        self.conv_dw = DWConv7x7(dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim)

    def forward(self, x):  # x: [B, H, W, C]
        # (depthwise output, the input again for the shortcut: their gradients meet in the depthwise backward kernel)
        h, shortcut = H.dwconv7x7_shortcut(x, self.conv_dw.weight, self.conv_dw.bias)
        h = self.norm(h)
        m = self.mlp
        return H.mlp(h, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, "gelu", residual=shortcut,
                     colscale=self.gamma)


class Downsample(nn.Sequential):
    """timm: nn.Sequential(LayerNorm2d(cin), Conv2d(cin, cout, 2, 2)) -> keys '0.*', '1.*'."""

    def __init__(self, cin, cout):
        super().__init__(LayerNorm(cin, eps=1e-6), PatchConv2x2(cin, cout))


class Stage(nn.Module):
    def __init__(self, cin, cout, depth, first):
        super().__init__()
        self.downsample = nn.Identity() if first else Downsample(cin, cout)
        self.blocks = nn.Sequential(*[ConvNeXtBlock(cout) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class Head(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = LayerNorm(dim, eps=1e-6)


class ConvNeXtTiny(nn.Module):
    """forward(img NCHW fp32) -> [B, 768]."""

    num_features = 768

    def __init__(self, in_chans=3):
        super().__init__()
        self.stem = nn.Sequential(StemConv4x4(in_chans, DIMS[0]), LayerNorm(DIMS[0], eps=1e-6))
        stages, prev = [], DIMS[0]
        for i, (depth, dim) in enumerate(zip(DEPTHS, DIMS)):
            stages.append(Stage(prev, dim, depth, first=(i == 0)))
            prev = dim
        self.stages = nn.Sequential(*stages)
        self.head = Head(DIMS[-1])
        self._init_weights(in_chans)

    def _init_weights(self, in_chans):
        # timm convnext: trunc_normal_(std=.02) on conv / linear weights, zero biases
        for m in self.modules():
            if isinstance(m, (Linear, DWConv7x7, PatchConv2x2)):
                _trunc_normal_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, StemConv4x4):
                w = torch.empty(m.out_chans, in_chans, 4, 4)
                _trunc_normal_(w)
                with torch.no_grad():
                    m.weight.copy_(m.to_kernel(w))
                nn.init.zeros_(m.bias)

    def forward(self, img):
        x = self.stem(img)  # [B, OH, OW, 96]
        x = self.stages(x)
        B, Hh, Ww, C = x.shape
        pooled = x.reshape(B, C) if Hh * Ww == 1 else H.avgpool_tokens(x.reshape(B, Hh * Ww, C))
        return self.head.norm(pooled)
