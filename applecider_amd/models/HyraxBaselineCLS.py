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
    reference; the masked pre-training `train_step` (per-band random masking + the three-term product
    loss, :226-284) is the next row of SURVEY §8f and is not on the MI355X path yet."""

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

    def train_step(self, batch):
        raise NotImplementedError("MPTModel.train_step (masked pre-training) is not built on the "
                                  "MI355X path yet; see DESIGN.md section 7")

    to_tensor = staticmethod(HyraxBaselineCLS.to_tensor)
