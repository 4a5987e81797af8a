"""SpectraNet (multi-scale 1-D CNN over 4096-bin spectra) on the MI355X path.

Drop-in for src/applecider/models/spectranet.py: `SpectraNetBlock(in_channels, out_channels,
kernel_sizes, use_ln, do_pool)`, `make_stage(...)`, `SpectraNet(config, data_sample)`,
`forward((flux, labels, redshifts))`, `train_step`, `to_tensor`; same state_dict keys/shapes
(`all_stages.{i}.{j}.convs.{n}.weight` [Cout, Cin, k], ...).

This branch is 92 % of the model's FLOPs.  Sequences are channels-last
[B, L, C]; each block is
  conv bank   3 parallel 'same' Conv1d as implicit GEMMs on the matrix cores, written straight into
              the channel-concatenated buffer (no torch.cat); stage 1 (Cin = 1, k up to 1021) uses
              the 8-phase Toeplitz form (hipops._ConvGroup1d)
  LN + GELU   one row kernel over the 3*Cout channels of each position
  1x1 conv    plain GEMM, then MaxPool1d(4) as a streaming kernel
BatchNorm stages (use_ln=False; off the default path, default_config.toml:105) run `ac_batchnorm_fwd/bwd`
(column moments + fused GELU); their statistics are per process.
"""

from __future__ import annotations

import torch
import torch.nn as nn

from .. import hipops as H
from ..hyrax_compat import hyrax_model
from ._layers import BatchNorm1d, Conv1dTap, Dropout, LayerNorm, Linear, Marker, PointConv1d


class SpectraNetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_sizes, use_ln=True, do_pool=False):
        super().__init__()
        self.do_pool, self.use_ln = do_pool, use_ln
        self.k = len(kernel_sizes)
        self.kernel_sizes = tuple(int(k) for k in kernel_sizes)
        norm_channels = out_channels * self.k
        self.convs = nn.ModuleList([Conv1dTap(in_channels, out_channels, k) for k in self.kernel_sizes])
        # use_ln=False: nn.BatchNorm1d over the channels (spectranet.py:21); statistics are per process
        # (no cross-rank synchronisation, as in the reference)
        self.norm = LayerNorm(norm_channels) if use_ln else BatchNorm1d(norm_channels)
        if do_pool:
            self.total_pooled_channels = norm_channels
            self.downsample = PointConv1d(norm_channels, out_channels)

    def _handover16(self, x) -> bool:
        if not (self.do_pool and H.bf16_operands()):
            return False
        n, k = self.downsample.cout, self.downsample.cin   # the 1x1 conv must take the bf16 path
        return n % 8 == 0 and k % 8 == 0 and H._big(x.shape[0] * x.shape[1], n, k)

    def forward(self, x):  # x: [B, L, Cin] channels-last
        if not self.use_ln:
            y = H.conv_group1d(x, self.kernel_sizes, [c.weight for c in self.convs], [c.bias for c in self.convs])
            y = self.norm(y, act="gelu")
            return H.maxpool4(self.downsample(y)) if self.do_pool else y
        # a pooled block feeds LN+GELU straight into the 1x1 conv: in bf16 mode that hand-over (and
        # the gradient coming back) is bf16 only — the [B, L, 3*Cout] fp32 tensors are never written
        cout = self.convs[0].weight.shape[0]
        if (self.do_pool and self.downsample.cout == cout
                and H.tail_covered(x.shape[0], x.shape[1], x.shape[2], cout, len(self.kernel_sizes))):
            # split-bf16 mode: LayerNorm + GELU + 1x1 conv + MaxPool(4) are one kernel behind the conv bank (and two in
            # the backward pass); nothing as wide as the concatenated channels is written in between
            return H.conv_group1d(x, self.kernel_sizes, [c.weight for c in self.convs], [c.bias for c in self.convs],
                                  ln=(self.norm.weight, self.norm.bias, self.norm.eps),
                                  tail=(self.downsample.weight, self.downsample.bias))
        y = H.conv_group1d(x, self.kernel_sizes, [c.weight for c in self.convs],
                           [c.bias for c in self.convs],
                           ln=(self.norm.weight, self.norm.bias, self.norm.eps),
                           out16_only=self._handover16(x))
        if self.do_pool:
            y = H.maxpool4(self.downsample(y))
        return y


def make_stage(in_channel, out_channel, depth, kernel_sizes, use_ln=True, do_pool=True):
    """One SpectraNet stage: `depth` blocks, pooling on the last (spectranet.py:44-82).
    Returns (nn.Sequential, number of kernels)."""
    k = len(kernel_sizes)
    blocks = [SpectraNetBlock(in_channels=in_channel if i == 0 else out_channel * k,
                              out_channels=out_channel, kernel_sizes=kernel_sizes, use_ln=use_ln,
                              do_pool=(do_pool if i == depth - 1 else False))
              for i in range(depth)]
    return nn.Sequential(*blocks), k


@hyrax_model
class SpectraNet(nn.Module):
    def __init__(self, config=None, data_sample=None):
        super().__init__()
        self.config = config
        sc = config["model"]["SpectraNet"]
        self.redshift = sc["redshift"]
        kernel_sizes_per_stage = sc["kernel_sizes_per_stage"]
        depths, use_ln_stages, channels = sc["depths"], sc["use_ln_stages"], sc["channels"]
        flat_dim, class_order = sc["flat_dim"], sc["class_order"]
        # the reference's chained `!=` (spectranet.py:108) only compares neighbours; check all
        if not (len(depths) == len(use_ln_stages) == len(channels) == len(kernel_sizes_per_stage)):
            raise ValueError(
                "depths, use_ln_stages, channels, and kernel_sizes_per_stage must be the same length.")
        self.stages, self.ks = [], []
        for i in range(len(depths)):
            stage, k = make_stage(in_channel=1 if i == 0 else channels[i - 1], out_channel=channels[i],
                                  depth=depths[i], kernel_sizes=kernel_sizes_per_stage[i],
                                  use_ln=use_ln_stages[i], do_pool=i != len(depths) - 1)
            self.stages.append(stage)
            self.ks.append(k)
        self.all_stages = nn.Sequential(*self.stages)
        head = nn.Sequential(Linear(flat_dim, 384), LayerNorm(384), Marker("gelu"), Dropout(0.5),
                             Linear(384, 1 if self.redshift else class_order))
        if self.redshift:
            self.regressor = head
        else:
            self.classifier = head

    def forward(self, batch):
        x, _, _ = batch  # flux [B, 1, L]
        B, Cin, L = x.shape
        if Cin != 1:
            raise ValueError("SpectraNet expects flux of shape [B, 1, L]")
        h = self.all_stages(x.reshape(B, L, 1))  # [B,1,L] and [B,L,1] share memory for Cin = 1
        z = H.global_max(h)  # adaptive_max_pool1d(., 1): [B, C]
        head = self.regressor if self.redshift else self.classifier
        z = head[3](head[1](head[0](z), act="gelu"))
        out = head[4](z)
        return out.squeeze(1) if self.redshift else out

    def train_step(self, batch):
        """Uses self.optimizer / self.criterion injected by Hyrax (spectranet.py:172-184); without
        Hyrax, `applecider_amd.training.attach_defaults(model)` sets its defaults, SGD(0.01, 0.9) +
        CrossEntropyLoss.  Labels arrive as int16 (to_tensor, spectranet.py:204)."""
        if not hasattr(self, "optimizer") or not hasattr(self, "criterion"):
            raise AttributeError("SpectraNet.train_step needs self.optimizer and self.criterion (Hyrax "
                                 "injects them; standalone: applecider_amd.training.attach_defaults(model))")
        _, labels, redshifts = batch
        self.optimizer.zero_grad()
        outputs = self(batch)
        loss = self.criterion(outputs, redshifts if self.redshift else labels)
        loss.backward()
        self.optimizer.step()
        return {"loss": loss.item()}

    @staticmethod
    def to_tensor(data_dict):
        """Sample dict -> (flux f32, label i16, redshift f32); numpy only (spectranet.py:186-206)."""
        import numpy as np

        if "data" not in data_dict:
            raise ValueError("Data dictionary must have a 'data' key.")
        data = data_dict["data"]
        return (np.asarray(data.get("flux", []), dtype=np.float32),
                np.asarray(data.get("label", []), dtype=np.int16),
                np.asarray(data.get("redshift", []), dtype=np.float32))
