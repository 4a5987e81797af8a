"""What Hyrax injects into a model before it calls `train_step` (the `@hyrax_model` decorator reads
`config["criterion"]` / `config["optimizer"]` and sets `self.criterion` / `self.optimizer`; log lines
docs/pre_executed/testing/spectranet_testing.ipynb cell 14: torch.optim.SGD(lr 0.01, momentum 0.9) +
torch.nn.CrossEntropyLoss).  Hyrax is out of tree and out of scope; this module provides the same two
attributes on the MI355X path so that `SpectraNet.train_step` (spectranet.py:172-184) runs standalone.
"""

from __future__ import annotations

import torch

from . import hipops as H
from .optim import FlatSGD


class CrossEntropyLoss:
    """torch.nn.CrossEntropyLoss() for class-index targets of any integer dtype (the reference's
    `to_tensor` emits int16 labels, spectranet.py:204) or soft float targets [B, C]."""

    def __call__(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if target.is_floating_point():
            return H.cross_entropy_soft(logits, target)
        return H.cross_entropy_index(logits, target)


class MSELoss:
    """torch.nn.MSELoss() (mean reduction): the criterion of SpectraNet's `redshift = True` configuration, whose
    train_step hands (outputs [B], redshifts [B]) to `self.criterion` (spectranet.py:178-179)."""

    def __call__(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return H.mse_loss(pred, target)


def attach_defaults(model, lr: float = 0.01, momentum: float = 0.9, weight_decay: float = 0.0):
    """Give `model` the optimizer / criterion pair Hyrax's defaults inject: SGD(lr, momentum) over the
    flat parameter buffer (one streaming kernel per step) and CrossEntropyLoss — MSELoss when the model is a
    regressor (`model.redshift`).  Returns the model."""
    model.optimizer = FlatSGD(model.parameters(), lr=lr, momentum=momentum, weight_decay=weight_decay)
    model.criterion = MSELoss() if getattr(model, "redshift", False) else CrossEntropyLoss()
    return model
