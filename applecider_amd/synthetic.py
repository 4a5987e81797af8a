"""Seeded synthetic batches with the sample-dict schema of src/applecider/datasets
(SURVEY.md §8d).  Host-side numpy only; used by bench.py, the tests and the golden generator.

Shapes: image f32[B,3,63,63] (triplet cutouts, alert_processor.py:20-51), metadata f32[B,24]
(5 % sentinel -999, preprocess_multimodal.py:720-728), photometry f32[B,L,7]
(dt, dt_prev, logf, logfe, one-hot band x3; pad_mask True beyond the event count,
photo_dataset.py:136-138), spectra f32[B,1,4096] (mean/MAD normalised, 10 % all-zero rows,
preprocess_multimodal.py:601-608,778-780), labels int64[B] in 0..4.
"""

from __future__ import annotations

import numpy as np


def make_batch(B: int, seed: int = 0, L: int = 128, spec_len: int = 4096, sentinels: bool = True,
               n_classes: int = 5, min_len: int = 8) -> dict:
    rng = np.random.default_rng(seed)
    image = rng.standard_normal((B, 3, 63, 63)).astype(np.float32)
    image -= np.median(image, axis=(2, 3), keepdims=True)
    metadata = rng.standard_normal((B, 24)).astype(np.float32)
    if sentinels:
        metadata[rng.random((B, 24)) < 0.05] = -999.0
    # photometry
    lens = rng.integers(min(min_len, L), L + 1, size=B)
    t = np.sort(rng.uniform(0, 100, size=(B, L)), axis=1)
    dt = np.log1p(t - t[:, :1])
    dt_prev = np.log1p(np.diff(t, axis=1, prepend=t[:, :1]))
    logf = rng.normal(1.5, 0.5, size=(B, L))
    logfe = np.abs(rng.normal(0.05, 0.02, size=(B, L)))
    cont = np.stack([dt, dt_prev, logf, logfe], -1)
    cont = (cont - cont.mean((0, 1))) / (cont.std((0, 1)) + 1e-8)
    band = np.eye(3)[rng.integers(0, 3, size=(B, L))]
    photometry = np.concatenate([cont, band], -1).astype(np.float32)
    pad_mask = np.arange(L)[None, :] >= lens[:, None]
    photometry[pad_mask] = 0.0
    # spectra: cumulative-sum "continuum", mean/MAD normalised, some missing
    spec = np.cumsum(rng.standard_normal((B, spec_len)), axis=1)
    spec -= spec.mean(1, keepdims=True)
    mad = np.median(np.abs(spec - np.median(spec, 1, keepdims=True)), 1, keepdims=True)
    spec = (spec / (mad + 1e-8)).astype(np.float32)
    spec[rng.random(B) < 0.10] = 0.0
    labels = rng.integers(0, n_classes, size=B).astype(np.int64)
    onehot = np.eye(n_classes, dtype=np.float32)[labels]
    return {
        "image": image, "metadata": metadata, "photometry": photometry, "pad_mask": pad_mask,
        "spectra": spec[:, None, :], "label": labels, "target": onehot, "lengths": lens,
    }
