"""AppleCider — the 4-modality fusion classifier (photometry + image/metadata + spectra).

The fusion module only exists in the reference's archive
(_archive/AppleCider/core/model.py:8-67, newer variant _archive/notebooks/brew_cider.py:807-862) and
is not importable there as written.  This module keeps the archive's constructor contract
`AppleCider(config)` with its key names (mode, p_d_model, p_n_heads, p_n_layers, p_dropout, max_len,
num_classes, hidden_dim, fusion) and its forward signature
`forward(photometry, photometry_mask, metadata, images, spectra)`.

Documented definition of the branch encoders: the archive's spectra encoder
emitted 256-d features while src/ SpectraNet emits `class_order` logits; here the three encoders are
the live src/ models unchanged —
    photometry_encoder   = HyraxBaselineCLS with mode != "photo"  -> [B, p_d_model]
    spectra_encoder      = SpectraNet                              -> [B, class_order]
    img_metadata_encoder = AstroMiNN                               -> [B, 5]
each projected to hidden_dim, L2-normalised, averaged or concatenated (order: photometry,
image+metadata, spectra, brew_cider.py:851-856) and classified by `fc`.
"""

from __future__ import annotations

import copy

import torch
import torch.nn as nn

from .. import hipops as H
from ..config import default_config
from ..optim import FlatAdam
from ._layers import Linear
from .astrominn import AstroMiNN
from .HyraxBaselineCLS import HyraxBaselineCLS
from .spectranet import SpectraNet


class AppleCider(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.classification = True if config["mode"] == "all" else False
        model_cfg = copy.deepcopy(config.get("model_config") or default_config())
        pc = model_cfg["model"]["HyraxBaselineCLS"]
        pc.update({"d_model": config["p_d_model"], "n_heads": config["p_n_heads"],
                   "n_layers": config["p_n_layers"], "dropout": config["p_dropout"],
                   "max_len": config.get("max_len"), "num_classes": config["num_classes"],
                   "mode": "all", "pretrained_weights_path_": False, "use_probabilities": False})
        model_cfg["model"]["AstroMiNN"]["use_probabilities"] = False
        self.photometry_encoder = HyraxBaselineCLS(model_cfg)
        self.spectra_encoder = SpectraNet(model_cfg)
        self.img_metadata_encoder = AstroMiNN(model_cfg)
        hidden = config["hidden_dim"]
        self.photometry_proj = Linear(config["p_d_model"], hidden)
        self.spectra_proj = Linear(model_cfg["model"]["SpectraNet"]["class_order"], hidden)
        self.img_metadata_proj = Linear(5, hidden)
        if self.classification:
            self.fusion = config["fusion"]
            in_features = hidden * 3 if self.fusion == "concat" else hidden
            self.fc = Linear(in_features, config["num_classes"])
        # legacy step: Adam(lr, betas, weight_decay) + CrossEntropyLoss (brew_cider.py:1211,1229)
        self.optimizer = FlatAdam([{"params": list(self.parameters())}], lr=config.get("lr", 1e-3),
                                  betas=(config.get("beta1", 0.9), config.get("beta2", 0.999)),
                                  weight_decay=config.get("weight_decay", 0.0))

    # The three encoders are independent until the fusion head.  Each runs on its own HIP stream
    # (forked from / joined to the caller's stream with events), so the small launches of the image
    # and photometry branches (108-workgroup products, row kernels) fill the CUs that the tails of the
    # spectra branch's large kernels leave idle.  autograd replays every backward node on the stream
    # of its forward and synchronises at stream boundaries, so backward overlaps the same way.
    branch_streams = True

    def _streams(self, device):
        st = getattr(self, "_branch_streams", None)
        if st is None or st[0].device != device:
            # (stream priorities were tried for the side branches: no measurable effect)
            st = [torch.cuda.Stream(device=device) for _ in range(2)]
            # parameters' AccumulateGrad nodes live on the stream of their first use; gradients now
            # arrive from the branch streams.  autograd orders that with stream waits (what we want)
            # and would warn about it on every step.
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
            self._branch_streams = st
            H.register_side_streams(st)
        return st

    def get_embeddings(self, photometry, photometry_mask, metadata, images, spectra):
        if not (self.branch_streams and spectra.is_cuda):
            p_emb = self.photometry_proj(self.photometry_encoder((photometry, photometry_mask, None)))
            s_emb = self.spectra_proj(self.spectra_encoder((spectra, None, None)))
            im_emb = self.img_metadata_proj(self.img_metadata_encoder((metadata, images, None)))
            return H.l2_normalize(p_emb), H.l2_normalize(im_emb), H.l2_normalize(s_emb)
        main = torch.cuda.current_stream(spectra.device)
        s_img, s_pho = self._streams(spectra.device)
        H.ensure_mirrors(self.optimizer.fp)          # bf16 weight copies are refreshed before the fork
        for st in (s_img, s_pho):
            st.wait_stream(main)
        with torch.cuda.stream(s_img):
            im_emb = H.l2_normalize(self.img_metadata_proj(self.img_metadata_encoder((metadata, images, None))))
        with torch.cuda.stream(s_pho):
            p_emb = H.l2_normalize(self.photometry_proj(self.photometry_encoder((photometry, photometry_mask, None))))
        s_emb = H.l2_normalize(self.spectra_proj(self.spectra_encoder((spectra, None, None))))
        for st, t in ((s_img, im_emb), (s_pho, p_emb)):
            main.wait_stream(st)
            t.record_stream(main)                    # allocated on a branch stream, consumed on `main`
        return p_emb, im_emb, s_emb

    def forward(self, photometry, photometry_mask, metadata, images, spectra):
        p_emb, im_emb, s_emb = self.get_embeddings(photometry, photometry_mask, metadata, images, spectra)
        if not self.classification:
            raise NotImplementedError
        if self.fusion == "concat":
            emb = H.cat_cols([p_emb, im_emb, s_emb])
        elif self.fusion == "avg":
            emb = H.add(H.add(p_emb, im_emb), s_emb, 1.0 / 3.0)
        else:
            raise NotImplementedError
        return self.fc(emb)

    def train_step(self, batch):
        """batch = (photometry, photo_mask, metadata, images, spectra, labels): the 6-tuple of the
        legacy collate (Time2Vec.py:38-45); order of ops as Trainer.train_epoch
        (_archive/AppleCider/core/trainer.py:156-188)."""
        photometry, mask, metadata, images, spectra, labels = batch
        self.optimizer.zero_grad()
        logits = self.forward(photometry, mask, metadata, images, spectra)
        loss = H.cross_entropy_index(logits, labels)
        loss.backward()
        self.optimizer.step()
        return {"loss": loss}  # device scalar: the caller decides when to sync
