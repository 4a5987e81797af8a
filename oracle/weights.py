"""Closed-form deterministic parameter fill (TEST INFRASTRUCTURE ONLY).

Both sides of every parity test (reference / oracle / HIP product) regenerate their weights
from (parameter name, shape) alone, so fixtures store outputs, never weights.
numpy's PCG64 stream seeded with crc32(name) is stable across numpy versions.
"""

import zlib

import numpy as np
import torch


def _fan_in(shape):
    if len(shape) <= 1:
        return max(int(shape[0]) if len(shape) else 1, 1)
    n = 1
    for s in shape[1:]:
        n *= int(s)
    return max(n, 1)


def fill_value(name: str, shape, salt: int = 0) -> torch.Tensor:
    rng = np.random.default_rng((zlib.crc32(name.encode()) + 7919 * salt) & 0xFFFFFFFF)
    shape = tuple(int(s) for s in shape)
    u = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":  # BatchNorm running variance: positive
        v = 1.0 + 0.5 * u
    elif leaf == "gamma":  # ConvNeXt layer scale: O(1) so that the blocks are exercised
        v = 0.5 + 0.25 * u
    elif leaf == "weight" and len(shape) == 1:  # LayerNorm scale
        v = 1.0 + 0.1 * u
    elif leaf in ("bias", "in_proj_bias", "b0", "b") or len(shape) <= 1:
        v = 0.1 * u if leaf != "w" else u  # Time2Vec frequencies w stay O(1)
        if leaf in ("w0",):
            v = u
    elif leaf == "cls_tok":
        v = 0.5 * u
    else:
        v = u * np.float32(np.sqrt(3.0 / _fan_in(shape)))  # unit-variance-preserving
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def closed_form_state_dict(shapes: dict, salt: int = 0) -> dict:
    """shapes: {param name: shape} -> {param name: tensor}."""
    return {k: fill_value(k, s, salt) for k, s in shapes.items()}


def fill_module_(module: torch.nn.Module, salt: int = 0):
    """In-place closed-form fill of every parameter/buffer of a torch module (by state_dict key)."""
    sd = module.state_dict()
    new = closed_form_state_dict({k: v.shape for k, v in sd.items() if v.dtype.is_floating_point},
                                 salt)
    module.load_state_dict({**sd, **new})
    return module
