"""hipGraph-captured inference of the fused classifier (BASELINE config 5).

The reference's inference loop (docs/pre_executed/astrominn_example.ipynb cells 10-12) runs the model
in `eval()` on one batch after another and emits one record `{"id": obj_id, "tensor": scores}` per
alert, with `use_probabilities` turning logits into softmax probabilities (astrominn.py:297-298,
HyraxBaselineCLS.py:84-85).  On the MI355X path one forward is ≈500 kernel launches from Python
(≈10 ms of host time whatever the batch size), so the forward is captured ONCE into a HIP graph and
replayed per batch:

  * static shapes: the batch is padded to the captured batch size (the padded rows are zeros with a
    fully masked light curve and are dropped from the result);
  * no host synchronisation anywhere in the forward (the top-2 MoE routing is a device kernel,
    `ac_moe_top2_fwd`; dropout is inactive in eval), so every launch is capturable;
  * the kernels receive torch's capture stream through `hipops._stream()`; buffers allocated during
    capture come from the graph's private pool and live as long as the object;
  * inputs are copied into static device buffers (one `copy_` each, H2D from pinned memory when the
    caller hands host tensors), outputs are read from a static buffer.

Weights are read through pointers at replay time: an optimizer step or `load_state_dict` between
replays is picked up (bf16 mode: call `refresh()` so the bf16 parameter mirrors are rebuilt).
"""

from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch

from . import hipops as H

_INPUT_KEYS = ("photometry", "pad_mask", "metadata", "image", "spectra")


class GraphedClassifier:
    """`model`: applecider_amd.models.applecider.AppleCider on its GPU.  `batch_size`: captured batch
    (BASELINE config 5 uses 2048).  `seq_len`: light-curve length of the captured shape."""

    def __init__(self, model, batch_size: int, seq_len: int = 128, spec_len: int = 4096,
                 use_probabilities: bool = True, warmup: int = 2):
        self.model = model.eval()
        self.B, self.L = int(batch_size), int(seq_len)
        self.use_probabilities = use_probabilities
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedClassifier needs the model on a GPU (no CPU fallback)")
        self.device = dev
        B = self.B
        self.static = {
            "photometry": torch.zeros(B, self.L, 7, device=dev),
            "pad_mask": torch.ones(B, self.L, device=dev, dtype=torch.bool),
            "metadata": torch.zeros(B, 24, device=dev),
            "image": torch.zeros(B, 3, 63, 63, device=dev),
            "spectra": torch.zeros(B, 1, spec_len, device=dev),
        }
        self.static["pad_mask"][:, 0] = False  # an all-padded row would be a 0/0 softmax: keep one key
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out: Optional[torch.Tensor] = None
        self._capture(warmup)

    def _forward(self):
        s = self.static
        logits = self.model(s["photometry"], s["pad_mask"], s["metadata"], s["image"], s["spectra"])
        return H.softmax_rows(logits) if self.use_probabilities else logits

    def _capture(self, warmup: int):
        # warm-up on a side stream: one-time work (hipFuncSetAttribute, offset tables, bf16 parameter
        # mirrors) must not land inside the capture
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self._forward()

    def refresh(self):
        """Rebuild what replays read besides the fp32 weights (bf16 parameter mirrors)."""
        opt = getattr(self.model, "optimizer", None)
        fp = getattr(opt, "fp", None)
        if fp is not None and fp.flat16 is not None:
            fp.refresh_mirrors()

    def eager(self, batch: dict) -> torch.Tensor:
        """The same forward without the graph (reference for tests and for odd shapes)."""
        with torch.no_grad():
            t = {k: torch.as_tensor(batch[k]).to(self.device) for k in _INPUT_KEYS}
            logits = self.model(t["photometry"], t["pad_mask"], t["metadata"], t["image"], t["spectra"])
            return H.softmax_rows(logits) if self.use_probabilities else logits

    def predict(self, batch: dict) -> torch.Tensor:
        """batch: dict with photometry [n,L,7], pad_mask [n,L] (True = padding), metadata [n,24],
        image [n,3,63,63], spectra [n,1,S]; n <= batch_size.  Returns scores [n, num_classes] (a view of
        the static output buffer: consume or clone before the next call)."""
        n = int(batch["metadata"].shape[0])
        if n > self.B:
            raise ValueError(f"batch of {n} exceeds the captured batch size {self.B}")
        if tuple(batch["photometry"].shape[1:]) != (self.L, 7):
            raise ValueError(f"photometry must be [n, {self.L}, 7] for this captured graph")
        for k in _INPUT_KEYS:
            src = torch.as_tensor(batch[k])
            self.static[k][:n].copy_(src, non_blocking=True)
        if n < self.B:   # neutral padding rows (their scores are dropped)
            self.static["pad_mask"][n:].fill_(True)
            self.static["pad_mask"][n:, 0] = False
            for k in ("photometry", "metadata", "image", "spectra"):
                self.static[k][n:].zero_()
        self.graph.replay()
        return self.out[:n]

    def records(self, batch: dict, ids: Sequence) -> List[dict]:
        """One `{"id", "tensor"}` record per alert, as the reference's inference results
        (astrominn_example.ipynb cell 12); one device-to-host copy per batch."""
        scores = self.predict(batch).cpu().numpy()
        if len(ids) != scores.shape[0]:
            raise ValueError("ids and batch differ in length")
        return [{"id": i, "tensor": scores[j]} for j, i in enumerate(ids)]

    def stream(self, batches: Iterable[dict]):
        """Generator over batches: yields a clone of the scores of each."""
        for b in batches:
            yield self.predict(b).clone()
