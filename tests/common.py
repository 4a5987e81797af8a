"""Shared helpers for the parity tests (CPU and GPU)."""

import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_SPECTRA = {"channels": [32, 32, 64, 64, 128], "flat_dim": 384}


def gold(name):
    return np.load(os.path.join(GOLD, name))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def compact(a, limit=65536, keep=4096):
    a = np.asarray(a)
    if a.size <= limit:
        return a
    return np.concatenate([a.reshape(-1)[:keep], [np.linalg.norm(a.astype(np.float64))]]).astype(np.float32)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


# Arithmetic-mode context of the golden model tests (tests/test_gpu_models.py `gmode`): every comparison made
# while a context is set is recorded as (test, mode, tensor) -> measured error, bound, and written to
# gpurun_out/parity_golden_modes.json (copied to profiles/ per round), so the benchmarked mode's errors against the
# reference's own goldens are REPORTED per tensor, not only bounded.
CONTEXT = {"test": None, "mode": None}
_RECORDS = {}


def set_context(test=None, mode=None):
    CONTEXT["test"], CONTEXT["mode"] = test, mode


def _record(name, e, tol):
    if CONTEXT["test"] is None:
        return
    import json
    _RECORDS.setdefault(CONTEXT["test"], {}).setdefault(CONTEXT["mode"], {})[name] = {"err": float(e), "bound": float(tol)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_golden_modes.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data.setdefault(CONTEXT["test"], {})[CONTEXT["mode"]] = _RECORDS[CONTEXT["test"]][CONTEXT["mode"]]
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def assert_close(a, b, tol, name="", x3=None):
    """max |a - b| / max |b| <= tol.  `x3`: the bound that applies instead in the benchmarked split-bf16 mode
    (CONTEXT mode "bf16x3") where it is stated to differ; unset = the same bound in both qualified modes."""
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    if torch.is_tensor(b):
        b = b.detach().cpu().numpy()
    if x3 is not None and CONTEXT["mode"] == "bf16x3":
        tol = x3
    e = relerr(a, b)
    _record(name, e, tol)
    assert e <= tol, f"{name}: rel-to-max error {e:.3e} > {tol:.1e}" + (f" [{CONTEXT['mode']}]" if CONTEXT["mode"] else "")


def cfg_default():
    from applecider_amd.config import default_config
    cfg = default_config()
    cfg["model"]["HyraxBaselineCLS"]["pretrained_weights_path_"] = False
    return cfg


def closed_form_sd(module, salt=0):
    """Closed-form weights for a product module, keyed/shaped like the reference state_dict."""
    from oracle.weights import closed_form_state_dict
    sd = module.state_dict()
    return closed_form_state_dict({k: v.shape for k, v in sd.items()}, salt)


def grads_by_ref_name(module):
    """Gradients of a product module converted to the reference's (checkpoint) layout."""
    from applecider_amd.models._layers import _LayoutLeaf
    from applecider_amd.models.Time2Vec import Time2Vec
    out = {}
    for mname, mod in module.named_modules():
        pre = mname + "." if mname else ""
        if isinstance(mod, Time2Vec):
            if mod.tw.grad is not None:
                out[pre + "w0"], out[pre + "w"] = mod.tw.grad[:1], mod.tw.grad[1:]
                out[pre + "b0"], out[pre + "b"] = mod.tb.grad[:1], mod.tb.grad[1:]
            continue
        for pname, p in mod.named_parameters(recurse=False):
            if p.grad is None:
                continue
            g = p.grad
            if isinstance(mod, _LayoutLeaf) and pname == "weight":
                g = mod.to_reference(g)
            out[pre + pname] = g
    return out
