"""Flat-buffer optimizers for the MI355X path.

All parameters of a model live in ONE contiguous fp32 buffer (and their gradients in a second
one): `param.data` / `param.grad` are views.  The optimizer step is then one streaming HIP kernel
per parameter group (ac_adam_flat / ac_sgd_flat) instead of hundreds of small launches, gradient
clipping is a single reduction, and data-parallel training all-reduces large contiguous slices of
the gradient buffer (applecider_amd/ddp.py) with no packing copies.

Semantics follow torch.optim.AdamW / Adam / SGD as the reference configures them
(astrominn.py:151-218, HyraxBaselineCLS.py:41,112, spectranet.py:172-184).
"""

from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import hipops as H
from ._lib import AdamSeg


class FlatParameters:
    """Re-homes the parameters of `groups` (list of {"params": [...], ...}) into flat buffers."""

    def __init__(self, groups: List[dict]):
        self.groups = groups
        self.params: List[torch.nn.Parameter] = []
        seen = set()
        for g in groups:
            g["params"] = [p for p in g["params"]]
            for p in g["params"]:
                if id(p) in seen:
                    raise ValueError("parameter appears in more than one group")
                seen.add(id(p))
                self.params.append(p)
        self.flat: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.flat16 = self.flatT16 = None
        self.flat_hi = self.flat_lo = None
        self.mirror_dirty, self.mirror_version = True, -1
        self.offsets: List[int] = []
        self.group_ranges: List[tuple] = []

    @staticmethod
    def _round(n: int, q: int = 64) -> int:  # 256-byte aligned starts
        return (n + q - 1) // q * q

    def is_current(self) -> bool:
        if self.flat is None or not self.params:
            return False
        first, last = self.params[0], self.params[-1]
        return (first.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[0]
                and last.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[-1]
                and first.grad is not None
                and first.grad.data_ptr() == self.grad.data_ptr() + 4 * self.offsets[0])

    def flatten(self):
        device = self.params[0].device
        total, offsets, ranges = 0, [], []
        for g in self.groups:
            begin = total
            for p in g["params"]:
                offsets.append(total)
                total += self._round(p.numel())
            ranges.append((begin, total))
        flat = torch.zeros(total, device=device, dtype=torch.float32)
        grad = torch.zeros(total, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(self.params, offsets):
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                p.grad = grad[off:off + n].view(p.shape)
        self.flat, self.grad, self.offsets, self.group_ranges = flat, grad, offsets, ranges
        self.flat16 = self.flatT16 = None
        self.flat_hi = self.flat_lo = None
        self.mirror_dirty, self.mirror_version = True, -1
        if flat.is_cuda:
            H.register_mirror(self)

    def refresh_mirrors(self):
        """16-bit operand copies of ALL parameters, made once per optimizer step.  bf16 / f16 modes: casts (and the
        transposes of the 2-D ones) in two launches.  Split-bf16 mode: the (hi, lo) planes of the whole buffer in ONE
        launch (ac_split_bf16) - every product that takes a weight as its B operand reads views of them
        (hipops.split16_w: the conv taps of SpectraNet, the plane-fed nn.Linear products)."""
        if H.x3_mode():
            if getattr(self, "flat_hi", None) is None or self.flat_hi.dtype != H._H16:
                self.flat_hi = torch.empty(self.flat.numel(), device=self.flat.device, dtype=H._H16)
                self.flat_lo = torch.empty_like(self.flat_hi)
            H.split16_into(self.flat, self.flat_hi, self.flat_lo)
            self.mirror_dirty, self.mirror_version = False, self.flat._version
            self.mirror_epoch = H._MATH_EPOCH
            self.mirror_pver = [p._version for p in self.params]
            return
        if self.flat16 is None or self.flat16.dtype != H._H16:
            self.flat16 = torch.empty(self.flat.numel(), device=self.flat.device, dtype=H._H16)
            self.flatT16 = torch.empty_like(self.flat16)
            segs, tiles = [], 0
            for p, off in zip(self.params, self.offsets):
                if p.dim() == 2:
                    r, c = p.shape
                    segs += [off, r, c, tiles]
                    tiles += -(-r // 64) * -(-c // 64)
            self._nseg, self._tiles = len(segs) // 4, tiles
            self._segs = (torch.tensor(segs, dtype=torch.int64, device=self.flat.device)
                          if segs else None)
        H.cast16_into(self.flat, self.flat16)
        if self._segs is not None:
            H.transpose_cast_segments(self.flat, self.flatT16, self._segs, self._nseg, self._tiles)
        self.mirror_dirty, self.mirror_version = False, self.flat._version
        self.mirror_epoch = H._MATH_EPOCH
        self.mirror_pver = [p._version for p in self.params]

    def ensure(self):
        if not self.is_current():
            self.flatten()
            return True
        return False

    def zero_grad(self):
        # gradients stay views of the flat buffer (set_to_none would break the layout)
        if self.grad is not None:
            self.grad.zero_()
            H.zero_pools_new_step()
            for p, off in zip(self.params, self.offsets):
                if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                    p.grad = self.grad[off:off + p.numel()].view(p.shape)


class FlatAdam:
    """Adam / AdamW over flat buffers.  `groups` use torch's keys: params, lr, betas, eps,
    weight_decay.  decoupled=True is AdamW."""

    def __init__(self, groups: Iterable[dict], lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, decoupled=False):
        groups = [dict(g) for g in groups]
        for g in groups:
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            g.setdefault("weight_decay", weight_decay)
        self.param_groups = groups
        self.fp = FlatParameters(groups)
        self.decoupled = decoupled
        self.step_count = 0
        self.exp_avg = self.exp_avg_sq = None
        # capturable=True (torch.optim's name for it): the step count also lives in HBM and the kernel
        # computes the bias corrections from it, so step() can sit inside a captured hipGraph
        self.capturable = False
        self.step_dev = None

    def set_capturable(self, on: bool = True):
        self._ensure()
        self.capturable = bool(on)
        if on:
            if self.step_dev is None:
                self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.fp.flat.device)
            self.step_dev.fill_(self.step_count)
        return self

    def _ensure(self):
        old = None
        if self.exp_avg is not None and self.step_count > 0 and not self.fp.is_current():
            # the parameters were re-homed (model.to(), p.data reassigned) after training started:
            # carry the moments over instead of silently restarting the optimizer
            old = self._moments_per_param()
        if self.fp.ensure() or self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.fp.flat)
            self.exp_avg_sq = torch.zeros_like(self.fp.flat)
            if old is None:
                self.step_count = 0
            else:
                self._load_moments(old)

    def _moments_per_param(self):
        out = []
        for p, off in zip(self.fp.params, self.fp.offsets):
            n = p.numel()
            out.append((self.exp_avg[off:off + n].detach().clone().view(p.shape),
                        self.exp_avg_sq[off:off + n].detach().clone().view(p.shape)))
        return out

    def _load_moments(self, moments):
        for (m, v), p, off in zip(moments, self.fp.params, self.fp.offsets):
            n = p.numel()
            self.exp_avg[off:off + n].copy_(m.reshape(-1).to(self.exp_avg.device))
            self.exp_avg_sq[off:off + n].copy_(v.reshape(-1).to(self.exp_avg.device))

    def state_dict(self):
        """torch.optim.Adam(W)-shaped state: {"state": {i: {"step", "exp_avg", "exp_avg_sq"}},
        "param_groups": [{..., "params": [indices]}]} with per-parameter tensors (copies), so that a
        checkpoint of the reference's optimizer and this one interchange."""
        self._ensure()
        state = {}
        for i, (m, v) in enumerate(self._moments_per_param()):
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m, "exp_avg_sq": v}
        groups, i = [], 0
        for g in self.param_groups:
            n = len(g["params"])
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": list(range(i, i + n))})
            i += n
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._ensure()
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for g, lg in zip(self.param_groups, sd["param_groups"]):
            if len(lg["params"]) != len(g["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size "
                                 "of optimizer's group")
            g.update({k: v for k, v in lg.items() if k != "params"})
        st = sd["state"]
        if st:
            self._load_moments([(st[i]["exp_avg"], st[i]["exp_avg_sq"]) for i in range(len(self.fp.params))])
            self.step_count = int(float(st[0]["step"]))
            if self.step_dev is not None:
                self.step_dev.fill_(self.step_count)

    def prepare(self):
        """Flatten now (call after the model sits on its GPU; train_step does it lazily)."""
        self._ensure()
        return self

    def zero_grad(self, set_to_none: bool = False):
        self._ensure()
        self.fp.zero_grad()

    def clip_grad_norm_(self, max_norm: float):
        """Device-side clip coefficient (no host sync); applied inside the next step()."""
        self._ensure()
        self._clip, self._sumsq = H.clip_coef(self.fp.grad, max_norm)
        return self._sumsq

    def step(self):
        self._ensure()
        self.step_count += 1
        segs = []
        for g, (b, e) in zip(self.param_groups, self.fp.group_ranges):
            segs.append(AdamSeg(b, e, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                float(g["eps"]), float(g["weight_decay"]), int(self.decoupled)))
        if self.capturable:
            H.step_advance(self.step_dev)
            H.adam_flat_dev(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, segs, self.step_dev,
                            getattr(self, "_clip", None))
        else:
            H.adam_flat(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, segs, self.step_count,
                        getattr(self, "_clip", None))
        self._clip = None
        self.fp.mirror_dirty = True
        H.clear_step_cache()  # cached bf16 weight copies are stale now

    @property
    def flat_grad(self):
        self._ensure()
        return self.fp.grad


class FlatSGD:
    """torch.optim.SGD(lr, momentum, weight_decay) over flat buffers."""

    def __init__(self, params, lr=0.01, momentum=0.0, weight_decay=0.0):
        self.param_groups = [{"params": list(params), "lr": lr, "momentum": momentum,
                              "weight_decay": weight_decay}]
        self.fp = FlatParameters(self.param_groups)
        self.buf = None
        self.first = True

    def _ensure(self):
        old = None
        if self.buf is not None and not self.first and not self.fp.is_current():
            old = [self.buf[off:off + p.numel()].detach().clone() for p, off in zip(self.fp.params, self.fp.offsets)]
        if self.fp.ensure() or self.buf is None:
            self.buf = torch.zeros_like(self.fp.flat)
            if old is None:
                self.first = True
            else:   # parameters re-homed after training started: keep the momentum
                for b, p, off in zip(old, self.fp.params, self.fp.offsets):
                    self.buf[off:off + p.numel()].copy_(b.to(self.buf.device))

    def state_dict(self):
        """torch.optim.SGD-shaped state ({"state": {i: {"momentum_buffer"}}, "param_groups": [...]})."""
        self._ensure()
        state = {} if self.first else {
            i: {"momentum_buffer": self.buf[off:off + p.numel()].detach().clone().view(p.shape)}
            for i, (p, off) in enumerate(zip(self.fp.params, self.fp.offsets))}
        g = self.param_groups[0]
        return {"state": state, "param_groups": [{**{k: v for k, v in g.items() if k != "params"},
                                                  "params": list(range(len(g["params"])))}]}

    def load_state_dict(self, sd):
        self._ensure()
        self.param_groups[0].update({k: v for k, v in sd["param_groups"][0].items() if k != "params"})
        if sd["state"]:
            for i, (p, off) in enumerate(zip(self.fp.params, self.fp.offsets)):
                self.buf[off:off + p.numel()].copy_(sd["state"][i]["momentum_buffer"].reshape(-1).to(self.buf.device))
            self.first = False

    def prepare(self):
        self._ensure()
        return self

    def zero_grad(self, set_to_none: bool = False):
        self._ensure()
        self.fp.zero_grad()

    def step(self):
        self._ensure()
        g = self.param_groups[0]
        H.sgd_flat(self.fp.flat, self.fp.grad, self.buf, float(g["lr"]), float(g["momentum"]),
                   float(g["weight_decay"]), self.first)
        self.first = False
        self.fp.mirror_dirty = True
        H.clear_step_cache()

    @property
    def flat_grad(self):
        self._ensure()
        return self.fp.grad
