"""Host-side collation of AppleCiDEr sample dicts and the pinned-host async H2D staging
(SURVEY.md §8f-1).

`collate_photometry` mirrors PhotoEventsDataset.collate (src/applecider/datasets/photo_dataset.py:
117-152): pad to max(257, longest) with zeros, build the True=padding mask, truncate to 257.
`collate_fused` mirrors the 5-modal legacy collate (src/applecider/models/Time2Vec.py:18-45) — tuple
order (photometry, photo_mask, metadata, images, spectra, labels) — but takes mean/std as arguments
(the reference hard-codes a path) and leaves device placement to `PinnedStager`, which replaces the
blocking `.to(device)` calls (brew_cider.py:991-994) with double-buffered pinned staging and
non-blocking copies on a dedicated copy stream.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

MAX_LEN = 257  # default_config.toml:64


def collate_photometry(batch: Sequence[dict]) -> dict:
    seqs, labels = [], []
    for item in batch:
        seqs.append(np.asarray(item["data"]["photometry"]))
        if "label" in item["data"]:
            labels.append(item["data"]["label"])
    lengths = [s.shape[0] for s in seqs]
    width = max(MAX_LEN, max(lengths))
    out = np.zeros((len(seqs), width, seqs[0].shape[1]), dtype=seqs[0].dtype)
    pad_mask = np.ones((len(seqs), width), dtype=bool)
    for i, s in enumerate(seqs):
        out[i, :s.shape[0]] = s
        pad_mask[i, :s.shape[0]] = False
    return {"data": {"photometry": out[:, :MAX_LEN, :], "label": np.array(labels),
                     "pad_mask": pad_mask[:, :MAX_LEN],
                     "mean": np.array(batch[0]["data"]["mean"]),
                     "std": np.array(batch[0]["data"]["std"])}}


def collate_fused(batch: Sequence[tuple], mean: np.ndarray, std: np.ndarray, max_len: int = None):
    """batch of (photo_seq [l,7], metadata [24], image [3,63,63], spectrum [1,S], label) ->
    host tuple (photometry f32[B,L,7] normalised, photo_mask bool[B,L], metadata, images, spectra,
    labels i64)."""
    photo, metadata, images, spectra, labels = zip(*batch)
    lens = [np.asarray(s).shape[0] for s in photo]
    L = max_len or max(lens)
    B = len(batch)
    ph = np.zeros((B, L, 7), dtype=np.float32)
    mask = np.ones((B, L), dtype=bool)
    for i, s in enumerate(photo):
        n = min(lens[i], L)
        ph[i, :n] = np.asarray(s, dtype=np.float32)[:n]
        mask[i, :n] = False
    ph[..., :4] = (ph[..., :4] - np.asarray(mean, np.float32)) / (np.asarray(std, np.float32) + 1e-8)
    return (ph, mask, np.stack([np.asarray(m, np.float32) for m in metadata]),
            np.stack([np.asarray(im, np.float32) for im in images]),
            np.stack([np.asarray(sp, np.float32) for sp in spectra]),
            np.asarray(labels, dtype=np.int64))


def pack_ragged_photometry(photo: Sequence, max_len: int = None):
    """Ragged light curves -> (flat f32 [sum, 7], offsets i64 [B], lens i32 [B], L): what the device-side collate takes.
    No padding and no arithmetic on the host; L = max_len or the longest curve."""
    arrs = [np.asarray(s, dtype=np.float32).reshape(-1, 7) for s in photo]
    lens = np.array([a.shape[0] for a in arrs], dtype=np.int32)
    offsets = np.zeros(len(arrs), dtype=np.int64)
    if len(arrs) > 1:
        offsets[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    flat = np.concatenate(arrs, 0) if arrs else np.zeros((0, 7), np.float32)
    if flat.shape[0] == 0:
        flat = np.zeros((1, 7), np.float32)      # (a batch of empty curves still needs a valid pointer)
    return flat, offsets, lens, int(max_len or lens.max())


def collate_fused_device(batch: Sequence[tuple], mean, std, stager: "PinnedStager", max_len: int = None):
    """collate_fused with the pad / truncate / standardise step ON THE DEVICE: the ragged photometry goes
    over PCIe as it is (flat rows + offsets + lengths through the pinned stager), ac_collate_photometry builds the padded,
    standardised [B, L, 7] tensor and the padding mask in HBM - bit-identical to the host arithmetic of `collate_fused`.
    Returns the device tuple (photometry, photo_mask bool, metadata, images, spectra, labels)."""
    from .. import _lib, hipops as H
    photo, metadata, images, spectra, labels = zip(*batch)
    flat, offsets, lens, L = pack_ragged_photometry(photo, max_len)
    host = (flat, offsets, lens, np.asarray(mean, np.float32).reshape(4), np.asarray(std, np.float32).reshape(4),
            np.stack([np.asarray(m, np.float32) for m in metadata]), np.stack([np.asarray(im, np.float32) for im in images]),
            np.stack([np.asarray(sp, np.float32) for sp in spectra]), np.asarray(labels, dtype=np.int64))
    d_flat, d_off, d_len, d_mean, d_std, d_meta, d_img, d_spec, d_lab = stager.stage(host)
    B = len(batch)
    out = torch.empty(B, L, 7, device=stager.device, dtype=torch.float32)
    mask = torch.empty(B, L, device=stager.device, dtype=torch.uint8)
    _lib.check(H._lib_().ac_collate_photometry(H._p(d_flat), H._p(d_off), H._p(d_len), H._p(d_mean), H._p(d_std), H._p(out),
                                               H._p(mask), B, L, 1, H._stream()), "ac_collate_photometry")
    for t in (d_flat, d_off, d_len, d_mean, d_std):      # consumed by a kernel on the current stream
        t.record_stream(torch.cuda.current_stream(stager.device))
    return out, mask.view(torch.bool), d_meta, d_img, d_spec, d_lab


class PinnedStager:
    """Double-buffered pinned-host staging + async H2D on a copy stream.

        stager = PinnedStager(device)
        dev_batch = stager.stage(host_tuple)          # copy, then make the current stream wait for it
    or, to overlap the copy of batch i+1 with the compute of batch i:
        ticket = stager.prefetch(host_tuple_next)     # copies are queued on the copy stream only
        ... launch step i ...
        dev_batch = stager.acquire(ticket)            # the current stream waits for the copy event

    The returned tensors are safe to use on the stream that called acquire()/stage(): the copy stream
    records an event after the last copy and that stream waits on it (no host synchronisation); the
    tensors are marked as used by it so the caching allocator does not hand their memory back to
    the copy stream early.  A slot's pinned buffers are rewritten only after the event of their
    previous copy has completed (host-side wait on that one event, `depth` batches later).
    Entries of `host_tuple` that already are pinned torch tensors (a loader that collates straight
    into pinned memory) are copied from directly.
    """

    def __init__(self, device, depth: int = 2):
        self.device = torch.device(device)
        self.depth = depth
        self.slot = 0
        self.host = [dict() for _ in range(depth)]
        self.events = [None] * depth
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.bytes_staged = 0

    def _pinned(self, slot, key, arr):
        if torch.is_tensor(arr):
            if arr.is_pinned():
                return arr
            t = arr.contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(arr))
        buf = self.host[slot].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=self.device.type == "cuda")
            self.host[slot][key] = buf
        buf.copy_(t)
        return buf

    def prefetch(self, host_tuple):
        """Queue the H2D copies of one batch on the copy stream; returns a ticket for acquire()."""
        slot = self.slot
        self.slot = (self.slot + 1) % self.depth
        if self.events[slot] is not None:
            self.events[slot].synchronize()  # pinned buffers of this slot are free again
        if self.copy_stream is None:
            return (None, tuple(a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
                                for a in host_tuple))
        outs = []
        with torch.cuda.stream(self.copy_stream):
            for i, a in enumerate(host_tuple):
                src = self._pinned(slot, i, a)
                self.bytes_staged += src.numel() * src.element_size()
                outs.append(src.to(self.device, non_blocking=True))
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.events[slot] = ev
        return (ev, tuple(outs))

    def acquire(self, ticket):
        """Order the current stream after the ticket's copies and hand out the device tensors."""
        ev, outs = ticket
        if ev is None:
            return outs
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for o in outs:
            o.record_stream(cur)
        return outs

    def stage(self, host_tuple):
        return self.acquire(self.prefetch(host_tuple))
