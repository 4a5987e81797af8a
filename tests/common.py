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


# ----------------------------------------------------------------------------- max-pool routing (gradient parity)
class routing_tap:
    """with routing_tap() as tap: ... product forward ...  -> tap.routing() = the oracle's `routing` argument holding
    the positions the PRODUCT's max-pools selected.  MaxPool1d / adaptive_max_pool1d make the gradient discontinuous:
    a window whose two largest values differ by less than the two implementations' rounding may send its gradient to
    the other position.  Evaluating the oracle under the product's routing removes exactly that effect, so the
    gradients can be held to the tight bound; `routing_mismatches` then bounds how many windows were routed differently
    and shows that each of them was a near-tie."""

    def __enter__(self):
        from applecider_amd import hipops as H
        self.H, self.items = H, []
        H.ROUTING_TAP = self.items
        return self

    def __exit__(self, *exc):
        self.H.ROUTING_TAP = None

    def relu_gates(self):
        """{oracle site: [gate masks in call order]} of the product's ReLUs: the encoder feed-forward (hidden width
        4 * d_model = 512, one per layer) and the image head's Linear(768 -> 384) + ReLU (astrominn.py:27)."""
        relus = [t.detach().cpu() > 0 for kind, t in self.items if kind == "relu"]
        return {"encoder_ff": [t for t in relus if t.shape[-1] == 512],
                "image_head": [t for t in relus if t.shape[-1] == 384]}

    def routing(self):
        pools = [idx.permute(0, 2, 1).cpu().long() for kind, idx in self.items if kind == "pool4"]
        glob = [idx.cpu().long() for kind, idx in self.items if kind == "global"]
        assert len(glob) == 1, "one SpectraNet forward per tap"
        return {"pool": pools, "global": glob[0]}


def routing_mismatches(routing, max_margin=1e-3, max_frac=2e-3):
    """After an oracle forward under `routing`: number of windows the product routed differently from the oracle's own
    arg-max, asserting every one of them is a near-tie (top-2 margin <= max_margin of the tensor's max-abs) and that
    they are few (<= max_frac of the windows)."""
    n_bad = n_all = n_tied = 0
    worst = 0.0
    pairs = list(zip(routing["pool"], routing["argmax"], routing["margins"]))
    pairs.append((routing["global"], routing["global_argmax"], routing["global_margin"]))
    for given, own, margin in pairs:
        diff = given != own
        n_all += diff.numel()
        if diff.any():
            # EXACT ties in the oracle (margin 0: the constant rows behind an all-zero spectrum, 10 % of the synthetic
            # batch - every position of such a row holds the same value and receives the same upstream gradient) are
            # counted apart: which of the equal positions a transform-domain convolution's rounding favours is free
            n_tied += int((diff & (margin == 0)).sum())
            n_bad += int((diff & (margin > 0)).sum())
            worst = max(worst, float(margin[diff].max()))
    assert worst <= max_margin, f"a max-pool window with a clear winner was routed differently (margin {worst:.2e})"
    assert n_bad <= max(2, max_frac * n_all), f"{n_bad} of {n_all} max-pool windows routed differently ({n_tied} exact ties apart)"
    routing["n_exact_ties_routed_differently"] = n_tied
    return n_bad, n_all, worst


def relu_gate_mismatches(gates, max_margin=1e-3, max_frac=1e-3):
    """After an oracle forward under oracle.functional.RELU_GATES = gates: gates that differ from the oracle's own,
    asserting each differing unit's pre-activation is within max_margin of zero (relative to the tensor's max-abs)
    and that they are few."""
    n_bad = n_all = 0
    worst = 0.0
    for site, seen in gates.get("_seen", {}).items():
        given = gates.get(site)
        if given is None:
            continue
        assert len(given) == len(seen), (site, len(given), len(seen))
        for g, (own, mag) in zip(given, seen):
            diff = g.reshape(own.shape) != own
            n_all += diff.numel()
            if diff.any():
                n_bad += int(diff.sum())
                worst = max(worst, float(mag[diff].max()))
    assert worst <= max_margin, f"a ReLU gate with a clear sign differs (|pre| / max = {worst:.2e})"
    assert n_bad <= max(4, max_frac * n_all), f"{n_bad} of {n_all} ReLU gates differ"
    return n_bad, n_all, worst
