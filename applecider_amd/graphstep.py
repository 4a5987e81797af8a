"""Whole training step as ONE hipGraph launch.

The eager step of the fused 4-modality model is ~1 200 kernel launches issued from Python (~25 ms of host
time at B = 512): once the kernels are faster than that, the host is the floor.  `GraphedTrainStep`
captures zero_grad -> forward -> loss -> backward -> optimizer step (three encoder streams included: the
fork / join events become graph edges) and replays it with one launch per step.

What makes the captured step a *training* step and not a replay of the same one:
  * dropout / attention-dropout / masking seeds: the host seeds baked into the captured launches are mixed
    on the device with a step counter in HBM (`hipops.enable_device_step`; the `step` argument of
    ac_dropout / ac_mha_* / ac_mpt_mask and `ac_gemm_desc.drop_step`), which the graph itself advances
    first thing every replay (`ac_step_advance`);
  * Adam's bias corrections: `FlatAdam.set_capturable()` keeps the step count in HBM as well
    (`ac_adam_flat_dev`), advanced inside the graph;
  * the batch: copied into static input tensors before each replay (device-to-device, or straight from the
    pinned staging buffers of `datasets.collate.PinnedStager`).
Hyper-parameters (lr, betas, weight decay, clip norm) ARE baked in: the step re-captures itself when a
parameter group changed (an LR scheduler stepping once per epoch costs one capture per epoch).

Reference: the loop being replaced is Trainer.train_epoch (_archive/AppleCider/core/trainer.py:156-188) /
Hyrax's per-batch `model.train_step(batch)` (astrominn.py:189-197); torch users get the same thing from
torch.cuda.graphs with `capturable=True` optimizers.

Single-GPU only for now: the bucketed gradient exchange (`ddp.GradBuckets`) is driven by autograd hooks in
Python and is not captured.
"""
from __future__ import annotations

import warnings
from typing import Callable, Optional, Sequence

import torch

from . import hipops as H


def _default_step(model, batch):
    out = model.train_step(batch)
    return out["loss"] if isinstance(out, dict) else out


def _accumulate_node(p):
    """The AccumulateGrad node of a leaf: autograd keeps ONE per parameter for as long as anything references
    it (a live autograd graph does) and builds a new one otherwise."""
    fn = p.view_as(p).grad_fn
    return fn.next_functions[0][0] if fn is not None and fn.next_functions else None


def stale_autograd_parameters(params, limit: int = 4):
    """Names / indices of parameters whose AccumulateGrad node is held alive by an autograd graph of an
    EARLIER step (a kept `loss`, a saved output with a grad_fn ...).  Such a node is bound to the stream its
    step ran on; replaying it inside a capture pulls that stream into the capture, which on ROCm 7.2 ends
    in a segfault inside hipStreamEndCapture rather than in an error.

    Detection does not depend on autograd's "AccumulateGrad node's stream" warning (the fused model
    silences it process-wide, models/applecider.py `_streams`): a token is written into the node's metadata
    dict, the Python reference is dropped, and the node is looked up again — the token is still there only if
    something else kept the C++ node alive."""
    token = object()
    stale = []
    for i, p in enumerate(params):
        if not (torch.is_tensor(p) and p.requires_grad and p.is_leaf):
            continue
        n = _accumulate_node(p)
        if n is None:
            continue
        n.metadata["_ac_capture_probe"] = token
        del n
        n = _accumulate_node(p)
        alive = n is not None and n.metadata.get("_ac_capture_probe") is token
        if n is not None:
            n.metadata.pop("_ac_capture_probe", None)
        del n
        if alive:
            stale.append(i)
            if len(stale) >= limit:
                break
    return stale


_STALE_MSG = ("GraphedTrainStep: an autograd graph of an earlier eager step is still alive (e.g. a kept `loss` "
              "tensor, parameters {idx}); drop it (`del loss` / `.detach()`) before building or re-capturing "
              "the graphed step")


def _hyper_signature(opt):
    sig = []
    for g in getattr(opt, "param_groups", []):
        sig.append(tuple((k, tuple(v) if isinstance(v, (list, tuple)) else v)
                         for k, v in sorted(g.items()) if k != "params"))
    return tuple(sig)


class GraphedTrainStep:
    """step = GraphedTrainStep(model, sample_batch); loss = step(batch) per iteration.

    model         a module of this package with `.optimizer` (FlatAdam / FlatSGD) already attached
    sample_batch  tuple of CUDA tensors with the shapes / dtypes of every later batch
    step_fn       (model, batch) -> loss tensor; default `model.train_step(batch)["loss"]`.  It must not
                  synchronise with the host (no .item()): AstroMiNN / SpectraNet's own train_step mirror the
                  reference's host-side loss bookkeeping, so pass a sync-free step for those.
    warmup        eager steps run on a side stream before the capture (allocations, lazy tables, the
                  flat parameter buffer).  With restore_state=True (default) parameters, optimizer state,
                  module buffers and the step counters are put back afterwards, so constructing the
                  object does not train.

    The returned loss is a static device scalar that the next call overwrites."""

    def __init__(self, model, sample_batch: Sequence[torch.Tensor], step_fn: Optional[Callable] = None,
                 warmup: int = 2, restore_state: bool = True):
        if H._grad_callbacks:
            raise RuntimeError("GraphedTrainStep is single-GPU: gradient buckets (ddp.GradBuckets) are attached "
                               "and their exchange is driven from Python hooks, which a graph does not replay")
        self.model = model
        self.opt = model.optimizer if hasattr(model, "optimizer") else model.this_optimizer
        self.step_fn = step_fn or _default_step
        self.warmup = int(warmup)
        self.static = tuple(t.clone() for t in sample_batch)
        if not all(t.is_cuda for t in self.static):
            raise ValueError("sample_batch must live on the GPU")
        self.opt.prepare()
        if hasattr(self.opt, "set_capturable"):
            self.opt.set_capturable(True)
        self.counter = H.enable_device_step(self.static[0].device)
        self.graph = None
        self.loss = None
        self.captures = 0
        self._capture(restore_state)

    # ------------------------------------------------------------------ state snapshot
    def _snapshot(self):
        opt = self.opt
        snap = {"flat": opt.fp.flat.clone(), "counter": self.counter.clone(),
                "buffers": [b.clone() for b in self.model.buffers()]}
        for name in ("exp_avg", "exp_avg_sq", "buf", "step_dev"):
            t = getattr(opt, name, None)
            if torch.is_tensor(t):
                snap[name] = t.clone()
        for name in ("step_count", "first"):
            if hasattr(opt, name):
                snap[name] = getattr(opt, name)
        return snap

    def _restore(self, snap):
        opt = self.opt
        opt.fp.flat.copy_(snap["flat"])
        self.counter.copy_(snap["counter"])
        for b, s in zip(self.model.buffers(), snap["buffers"]):
            b.copy_(s)
        for name in ("exp_avg", "exp_avg_sq", "buf", "step_dev"):
            if name in snap:
                getattr(opt, name).copy_(snap[name])
        if "step_count" in snap:
            opt.step_count = snap["step_count"]
        # FlatSGD.first stays False: the captured launch carries first_step = 0, and a zero momentum
        # buffer makes that identical to torch's first step (buf = grad)
        opt.fp.mirror_dirty = True
        H.clear_step_cache()

    # ------------------------------------------------------------------ capture
    def _one(self):
        H.step_advance()            # dropout step counter: first node of the graph
        return self.step_fn(self.model, self.static)

    def _capture(self, restore_state=True):
        self._check_no_stale_graph()
        snap = self._snapshot() if restore_state else None
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.stream(side):
                for _ in range(max(self.warmup, 1)):
                    self._one()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        for w in caught:
            if "AccumulateGrad node's stream" in str(w.message):
                # a loss (or any tensor with a grad_fn) of an EARLIER eager step is still referenced: it keeps
                # that step's AccumulateGrad nodes alive, which are bound to the stream of that step and
                # would pull the default stream into the capture (on ROCm 7.2 that ends in a segfault
                # inside hipStreamEndCapture, not in an error)
                raise RuntimeError(_STALE_MSG.format(idx="reported by autograd"))
            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        self._check_no_stale_graph()    # the warm-up's own graphs are gone (its losses were dropped)
        loss = self._record()
        if not (torch.is_tensor(loss) and loss.is_cuda):
            raise TypeError("step_fn must return the loss as a CUDA tensor (no host sync inside a captured step)")
        torch.cuda.synchronize()
        # the capture pass went through the host side of opt.step() once without executing anything
        if hasattr(self.opt, "step_count"):
            self.opt.step_count -= 1
        if snap is not None:
            self._restore(snap)
        self.graph, self.loss = self._graph_new, loss.detach()   # same storage, no autograd graph kept alive
        self._graph_new = None
        self._sig = _hyper_signature(self.opt)
        self.captures += 1
        self._invalidate_weight_copies()

    def _check_no_stale_graph(self):
        stale = stale_autograd_parameters(list(self.model.parameters()))
        if stale:
            raise RuntimeError(_STALE_MSG.format(idx=stale))

    def _record(self):
        """The capture itself (a separate method so that a test can stand in for it: a guard that misses
        must fail an assertion there, not reach hipStreamEndCapture)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = self._one()
        self._graph_new = g
        return loss

    # ------------------------------------------------------------------ replay
    def load(self, batch: Sequence[torch.Tensor]):
        """Copies one batch into the static inputs (non_blocking: pinned host tensors overlap)."""
        if len(batch) != len(self.static):
            raise ValueError(f"batch has {len(batch)} tensors, the captured step takes {len(self.static)}")
        for dst, src in zip(self.static, batch):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"batch tensor {tuple(src.shape)} {src.dtype} does not match the captured "
                                 f"{tuple(dst.shape)} {dst.dtype}: build another GraphedTrainStep for this shape")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)

    def __call__(self, batch: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        if _hyper_signature(self.opt) != self._sig:
            self._capture(restore_state=True)   # lr / betas / weight decay changed: baked values are stale
        if batch is not None:
            self.load(batch)
        self.graph.replay()
        if hasattr(self.opt, "step_count"):
            self.opt.step_count += 1
        self._invalidate_weight_copies()
        return self.loss

    def _invalidate_weight_copies(self):
        """The captured optimizer updates the parameters through raw pointers: no `_version` bump tells the
        host-side weight caches (16-bit mirrors, the (hi, lo) conv planes of bf16x3 mode) that they are stale.
        Eager `step()` marks them itself; a replay must too, or an eager forward between replays (validation
        after a graphed epoch) would keep using the copies of its first call."""
        self.opt.fp.mirror_dirty = True
        H.clear_step_cache()
