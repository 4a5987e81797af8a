"""HyraxBaselineCLS (photometry transformer) and FocalLoss on the MI355X path.

Drop-in for src/applecider/models/HyraxBaselineCLS.py: `HyraxBaselineCLS(config, data_sample)`,
`FocalLoss(gamma, alpha, eps, reduction)`, `forward((data, pad, labels))`, `train_step`,
`to_tensor`; state_dict keys are the reference's (including the constructed-but-unused `head`,
HyraxBaselineCLS.py:35).  Kernels: fused embed (in_proj + Time2Vec + CLS), per-(sample, head)
padded-mask attention, MFMA projections with fused residual / ReLU epilogues, post-LN rows,
fused focal loss forward+backward, one Adam kernel with device-side gradient clipping.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import hipops as H
from ..hyrax_compat import hyrax_model
from ..optim import FlatAdam
from ._layers import InProj8, LayerNorm, Linear
from .Time2Vec import Encoder, Time2Vec, embed_tokens, extend_pad_mask


class FocalLoss(nn.Module):
    def __init__(self, gamma: float = 2.0, alpha: torch.Tensor = None, eps: float = 0,
                 reduction: str = "mean"):
        super().__init__()
        self.gamma, self.alpha, self.eps, self.reduction = gamma, alpha, eps, reduction

    def forward(self, logits: torch.Tensor, target: torch.Tensor):
        alpha = self.alpha
        if alpha is not None:
            alpha = alpha.to(device=logits.device, dtype=torch.float32).contiguous()
        loss = H.focal_loss(logits, target, self.gamma, alpha, self.eps)
        if self.reduction == "mean":
            return loss
        return H.add(loss.reshape(1), loss.reshape(1), 0.5 * logits.shape[0]).reshape(())  # sum


@hyrax_model
class HyraxBaselineCLS(nn.Module):
    def __init__(self, config, data_sample=None):
        super().__init__()
        self.config = config
        self.criterion = FocalLoss()
        mc = config["model"]["HyraxBaselineCLS"]
        d = mc["d_model"]
        self.in_proj = InProj8(7, d)
        self.cls_tok = nn.Parameter(torch.zeros(1, 1, d))
        self.time2vec = Time2Vec(d)
        self.encoder = Encoder(d, mc["n_heads"], d * 4, mc["dropout"], mc["n_layers"])
        self.norm = LayerNorm(d)
        self.head = Linear(d, mc["num_classes"])
        self.classification = True if mc["mode"] == "photo" else False
        if self.classification:
            self.fc = Linear(d, mc["num_classes"])
        self.optimizer = FlatAdam([{"params": list(self.parameters())}], lr=1e-4)
        path = mc["pretrained_weights_path_"]
        if path:
            state_dict = torch.load(path, map_location="cpu")
            self.load_state_dict(state_dict, strict=False)
            print(f"Loaded pretrained weights from {path}")

    def encode(self, data, pad):
        h = embed_tokens(self.in_proj, self.time2vec, self.cls_tok, data)
        return self.encoder(h, extend_pad_mask(pad))

    def forward(self, x):
        data, pad, _ = x
        z = self.encode(data, pad)
        output = self.norm(H.take_token(z, 0))
        if self.classification:
            output = self.fc(output)
        if self.config["model"]["HyraxBaselineCLS"]["use_probabilities"]:
            output = H.softmax_rows(output)
        return output

    def train_step(self, batch):
        _, _, labels = batch
        decoded = self.forward(batch)
        loss = self.criterion(decoded, labels)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.clip_grad_norm_(1.0)  # coefficient stays on the device
        self.optimizer.step()
        return {"loss": loss.item(), "num_tdes": np.sum([labels.cpu().numpy() == 4])}

    @staticmethod
    def to_tensor(data_dict):
        """Sample dict -> (photometry, pad_mask, label); numpy only, same contract (and the same
        in-place normalisation of the first 4 channels) as HyraxBaselineCLS.py:122-166."""
        import numpy as np

        if "data" not in data_dict:
            raise ValueError("Data dictionary must contain 'data' key.")
        data = data_dict["data"]
        photo_tensor = data["photometry"]
        label_tensor = np.asarray(data.get("label", []), dtype=np.int64)
        photo_tensor[..., :4] = (photo_tensor[..., :4] - data["mean"]) / (data["std"] + 1e-8)
        if "pad_mask" in data.keys():
            return (photo_tensor, data["pad_mask"], label_tensor)
        false_mask = np.zeros((photo_tensor.shape[0], photo_tensor.shape[1] + 1), dtype=bool)
        return (photo_tensor, false_mask, label_tensor)


@hyrax_model
class MPTModel(nn.Module):
    """Masked-pretraining model (HyraxBaselineCLS.py:194-364): same encoder stack plus flux / band /
    dt heads.  Constructor, `forward(z)` (the three heads), `to_tensor` and the state_dict follow the
    reference; `train_step` is the masked pre-training step (:226-284): per-band random masking on the
    device (`ac_mpt_mask`), the encoder kernels of HyraxBaselineCLS, and the three-term product loss
    with its gradients in one pass (`ac_mpt_loss_fwd_bwd`)."""

    def __init__(self, config, data_sample=None):
        super().__init__()
        self.config = config
        mc = config["model"]["HyraxBaselineCLS"]
        d = mc["d_model"]
        self.encoder = Encoder(d, mc["n_heads"], d * 4, mc["dropout"], mc["n_layers"])
        self.in_proj = InProj8(7, d)
        self.cls_tok = nn.Parameter(torch.zeros(1, 1, d))
        self.time2vec = Time2Vec(d)
        self.head_flux = Linear(d, 1)
        self.head_band = Linear(d, 3)
        self.head_dt = Linear(d, 1)
        self.optimizer = FlatAdam([{"params": list(self.parameters())}], lr=1e-4, weight_decay=1e-2,
                                  decoupled=True)  # torch.optim.AdamW(lr=1e-4) defaults

    def forward(self, z):
        return self.head_flux(z), self.head_band(z), self.head_dt(z)

    def encode(self, data, pad):
        h = embed_tokens(self.in_proj, self.time2vec, self.cls_tok, data)
        return self.encoder(h, extend_pad_mask(pad))

    def _mask_batch(self, x, pad_mask, seed=None):
        """Per-band random masking on the device (HyraxBaselineCLS.py:286-319): modifies `x` in place
        and returns the selection, as the reference's method does."""
        return H.mpt_mask(x, pad_mask, self.config["model"]["HyraxBaselineCLS"]["mask_p"], seed)

    def pretrain_loss(self, data, pad, masked):
        """Forward of the pre-training step for an already masked batch (:232-278)."""
        mc = self.config["model"]["HyraxBaselineCLS"]
        B, L, _ = data.shape
        x8 = H.pad_channels(data, 8)
        zero_d = torch.zeros(self.time2vec.d_model, device=data.device)
        # [cls ; in_proj(x)] and [0 ; time2vec(t)] as two embed launches: the reference drops out the
        # time embedding alone (F.dropout(te, p), always active in this step, :243)
        emb = H.embed(x8, self.in_proj.weight, self.in_proj.bias, zero_d, zero_d, self.cls_tok.reshape(-1))
        te = H.embed(x8, torch.zeros_like(self.in_proj.weight), zero_d, self.time2vec.tw,
                     self.time2vec.tb, zero_d)
        h = H.add(emb, H.dropout(te, mc["dropout"], True))
        z = self.encoder(h, extend_pad_mask(pad))              # [B, L+1, d]
        f_hat, b_hat, dt_hat = self.head_flux(z), self.head_band(z), self.head_dt(z)
        return H.mpt_loss(f_hat, b_hat, dt_hat, data, masked,
                          (mc["lambda_f"], mc["lambda_b"], mc["lambda_dt"]))

    def train_step(self, batch):
        """Masked pre-training step (:226-284): mask -> encode -> three heads -> product loss ->
        backward -> clip_grad_norm_(1.0) -> AdamW.  `batch[0]` is modified in place by the masking,
        as in the reference."""
        data, pad = batch[0], batch[1]
        masked = self._mask_batch(data, pad)
        self.optimizer.zero_grad()
        loss = self.pretrain_loss(data, pad, masked)
        loss.backward()
        self.optimizer.clip_grad_norm_(1.0)
        self.optimizer.step()
        return {"loss": loss.item()}

    to_tensor = staticmethod(HyraxBaselineCLS.to_tensor)
