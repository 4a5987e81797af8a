"""Configuration the model constructors read.

The reference's ctors take the nested dict Hyrax builds from `default_config.toml`
(src/applecider/default_config.toml:1-119): ``config["model"][ClassName][key]``.
`default_config()` mirrors the [model.*] tables of that file; `load_toml()` is a small TOML-subset
reader (tables, scalars, possibly multi-line arrays) because Python 3.10 has no tomllib.
"""

from __future__ import annotations

import ast
import copy
import re

_DEFAULT = {
    "model": {
        "AstroMiNN": {
            "num_classes": 9, "num_mlp_experts": 4, "use_probabilities": False,
            "towers_hidden_dims": 16, "towers_outdims": 32,
            "fusion_hidden_dims": 128, "fusion_router_dims": 128, "fusion_outdims": 32,
            "cnn_lr": 2, "cnn_decay": 5e-2, "psf_lr": 0.5, "psf_decay": 5e-2,
            "mag_lr": 2, "mag_decay": 0.0, "lc_lr": 2, "lc_decay": 0.05,
            "spatial_lr": 2, "spatial_decay": 0.0, "coord_lr": 0.5, "coord_decay": 0.0,
            "nst1_lr": 2, "nst1_decay": 0.0, "nst2_lr": 2, "nst2_decay": 0.0,
            "fusion_lr": 1, "fusion_decay": 1e-2, "fusion_beta1": 0.9, "fusion_beta2": 0.999,
            "router_decay": 0.0, "router_lr": 1.5, "router_beta1": 0.9, "router_beta2": 0.999,
            "router_lr_2": 1, "router_beta1_2": 0.95, "router_beta2_2": 0.99,
            "beta1": 0.9, "beta2": 0.999, "eps": 5e-10,
        },
        "HyraxBaselineCLS": {
            "num_classes": 5, "pad_mask": 1, "mode": "photo",
            "d_model": 128, "n_heads": 8, "n_layers": 4, "dropout": 0.40, "max_len": 257,
            "lr": 5e-6, "weight_decay": 1e-2, "focal_gamma": 2.0,
            "cut_time_p": False, "p_dropout": 0.1, "jitter_scale": 0.10, "flux_nu": 8,
            "epochs": 150, "patience": 30, "seed": 42, "use_probabilities": False,
            "pretrained_weights_path_": "./pretrained_weights.pth",
            "lambda_f": 5.0, "lambda_b": 3.0, "lambda_dt": 5.0, "mask_p": 0.30,
        },
        "SpectraNet": {
            "redshift": False,
            "use_ln_stages": [True, True, True, True, True],
            "depths": [1, 1, 1, 1, 1],
            "channels": [64, 128, 256, 512, 1024],
            "kernel_sizes_per_stage": [[3, 61, 1021], [3, 31, 251], [3, 15, 61], [3, 11, 31],
                                       [3, 7, 13]],
            "class_order": 9, "flat_dim": 3072,
        },
    }
}


def default_config() -> dict:
    return copy.deepcopy(_DEFAULT)


def _parse_value(text: str):
    t = text.strip()
    t = re.sub(r"\btrue\b", "True", t)
    t = re.sub(r"\bfalse\b", "False", t)
    return ast.literal_eval(t)


def _strip_comment(line: str) -> str:
    out, in_s, q = [], False, ""
    for ch in line:
        if in_s:
            out.append(ch)
            if ch == q:
                in_s = False
        elif ch in "\"'":
            in_s, q = True, ch
            out.append(ch)
        elif ch == "#":
            break
        else:
            out.append(ch)
    return "".join(out).rstrip()


def _split_table(header: str):
    parts, cur, in_s, q = [], "", False, ""
    for ch in header:
        if in_s:
            if ch == q:
                in_s = False
            else:
                cur += ch
        elif ch in "\"'":
            in_s, q = True, ch
        elif ch == ".":
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return parts


def load_toml(path: str) -> dict:
    root: dict = {}
    table = root
    pending_key, pending = None, ""
    with open(path) as f:
        for raw in f:
            line = _strip_comment(raw)
            if pending_key is not None:
                pending += " " + line
                if pending.count("[") == pending.count("]"):
                    table[pending_key] = _parse_value(pending)
                    pending_key, pending = None, ""
                continue
            if not line.strip():
                continue
            s = line.strip()
            if s.startswith("[") and s.endswith("]") and "=" not in s:
                table = root
                for part in _split_table(s[1:-1]):
                    table = table.setdefault(part, {})
                continue
            key, _, val = s.partition("=")
            key = key.strip().strip("\"'")
            if val.count("[") != val.count("]"):
                pending_key, pending = key, val
            else:
                table[key] = _parse_value(val)
    return root
