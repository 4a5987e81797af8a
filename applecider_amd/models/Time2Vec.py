"""Time2Vec and the non-Hyrax BaselineCLS on the MI355X path (src/applecider/models/Time2Vec.py).

`Time2Vec(d_model)` keeps the reference's state_dict (w0[1], b0[1], w[d-1], b[d-1],
Time2Vec.py:55-60) but stores the frequencies/phases as two length-d vectors so that the fused
embedding kernel (in_proj + Time2Vec + CLS, ac_embed_fwd) reads them with one pointer each.
`collate` mirrors Time2Vec.py:18-45 (5-modal tuple order) but takes the normalisation statistics as
arguments instead of a hard-coded path and stages the batch through pinned host memory
(applecider_amd.datasets.collate).
"""

from __future__ import annotations

import torch
import torch.nn as nn

from .. import hipops as H
from ._layers import InProj8, LayerNorm, Linear


class Time2Vec(nn.Module):
    """v0 = w0*t + b0 ; v[i] = sin(w[i]*t + b[i])  (Time2Vec.py:48-72)."""

    def __init__(self, d_model):
        super().__init__()
        self.d_model = d_model
        tw = torch.randn(d_model)
        self.tw = nn.Parameter(tw)                 # [w0 | w]
        self.tb = nn.Parameter(torch.zeros(d_model))  # [b0 | b]

    # checkpoint layout = the reference's four tensors
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        tw = self.tw if keep_vars else self.tw.detach()
        tb = self.tb if keep_vars else self.tb.detach()
        destination[prefix + "w0"] = tw[:1]
        destination[prefix + "b0"] = tb[:1]
        destination[prefix + "w"] = tw[1:]
        destination[prefix + "b"] = tb[1:]

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        keys = [prefix + k for k in ("w0", "b0", "w", "b")]
        present = [k in state_dict for k in keys]
        if all(present):
            with torch.no_grad():
                self.tw.copy_(torch.cat([state_dict[keys[0]].reshape(1), state_dict[keys[2]].reshape(-1)]))
                self.tb.copy_(torch.cat([state_dict[keys[1]].reshape(1), state_dict[keys[3]].reshape(-1)]))
        elif strict:
            missing_keys.extend(k for k, p in zip(keys, present) if not p)

    def forward(self, t):
        """t: (B, L) -> (B, L, d_model).  Stand-alone use; the encoder uses the fused embed kernel."""
        B, L = t.shape
        x8 = torch.zeros(B, L, 8, device=t.device, dtype=torch.float32)
        x8[..., 0] = t
        zeros_w = torch.zeros(self.d_model, 8, device=t.device)
        zeros_b = torch.zeros(self.d_model, device=t.device)
        h = H.embed(x8, zeros_w, zeros_b, self.tw, self.tb, zeros_b)
        return h[:, 1:, :]


class EncoderLayer(nn.Module):
    """nn.TransformerEncoderLayer(d, h, ff, dropout, batch_first=True) defaults: post-LN, ReLU,
    eps 1e-5; parameter names as torch's (self_attn.in_proj_weight, ...)."""

    class _SelfAttn(nn.Module):
        def __init__(self, d):
            super().__init__()
            self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
            self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
            self.out_proj = Linear(d, d)
            nn.init.xavier_uniform_(self.in_proj_weight)
            nn.init.zeros_(self.out_proj.bias)

    def __init__(self, d_model, n_heads, dim_ff, dropout):
        super().__init__()
        self.n_heads, self.p = n_heads, float(dropout)
        self.self_attn = self._SelfAttn(d_model)
        self.linear1 = Linear(d_model, dim_ff)
        self.linear2 = Linear(dim_ff, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)

    def forward(self, x, pad_u8):
        sa = self.self_attn
        drop = self.training and self.p > 0
        qkv = H.linear(x, sa.in_proj_weight, sa.in_proj_bias)
        a = H.mha(qkv, pad_u8, self.n_heads, self.p, self.training)
        # x + dropout1(out_proj(attention)): one product with dropout and residual in its epilogue
        x = H.linear(a, sa.out_proj.weight, sa.out_proj.bias, residual=x, drop_p=self.p if drop else 0.0)
        x = self.norm1(x)
        l1, l2 = self.linear1, self.linear2
        x = H.mlp(x, l1.weight, l1.bias, l2.weight, l2.bias, "relu", p1=self.p, p2=self.p,
                  training=self.training, residual=x)
        return self.norm2(x)


class Encoder(nn.Module):
    """nn.TransformerEncoder(enc_layer, n_layers): keys 'layers.{i}.*'."""

    def __init__(self, d_model, n_heads, dim_ff, dropout, n_layers):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayer(d_model, n_heads, dim_ff, dropout)
                                     for _ in range(n_layers)])

    def forward(self, x, pad_u8):
        for layer in self.layers:
            x = layer(x, pad_u8)
        return x


def embed_tokens(in_proj: InProj8, time2vec: Time2Vec, cls_tok, data):
    """CLS + in_proj(x) + time2vec(x[...,0]) in one kernel; data is (B, L, 7)."""
    x8 = H.pad_channels(data, 8)
    return H.embed(x8, in_proj.weight, in_proj.bias, time2vec.tw, time2vec.tb, cls_tok)   # (the parameters themselves: gradient sinks)


def extend_pad_mask(pad):
    """F.pad(pad, (1, 0), False) as uint8 [B, L+1] (CLS is never masked)."""
    B, L = pad.shape
    out = torch.zeros(B, L + 1, device=pad.device, dtype=torch.uint8)
    out[:, 1:] = pad.to(torch.uint8)
    return out


class BaselineCLS(nn.Module):
    """Transformer encoder + class token (Time2Vec.py:80-124)."""

    def __init__(self, d_model, n_heads, n_layers, num_classes, dropout, max_len=None):
        super().__init__()
        self.in_proj = InProj8(7, d_model)
        self.cls_tok = nn.Parameter(torch.zeros(1, 1, d_model))
        self.time2vec = Time2Vec(d_model)
        self.encoder = Encoder(d_model, n_heads, d_model * 4, dropout, n_layers)
        self.norm = LayerNorm(d_model)
        self.head = Linear(d_model, num_classes)

    def forward(self, x, pad_mask):
        h = embed_tokens(self.in_proj, self.time2vec, self.cls_tok, x)
        z = self.encoder(h, extend_pad_mask(pad_mask))
        return self.head(self.norm(H.take_token(z, 0)))


class MPTModel(nn.Module):
    """Masked-pretraining heads over a BaselineCLS encoder (Time2Vec.py:128-142)."""

    def __init__(self, base_enc):
        super().__init__()
        self.encoder = base_enc.encoder
        d = base_enc.in_proj.out_features
        self.head_flux = Linear(d, 1)
        self.head_band = Linear(d, 3)
        self.head_dt = Linear(d, 1)

    def forward(self, z):
        return self.head_flux(z), self.head_band(z), self.head_dt(z)
