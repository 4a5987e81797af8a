"""CPU baseline leg of bench.py (TEST/BENCH INFRASTRUCTURE ONLY): times the oracle — the CPU
restatement pinned to the reference by tests/golden — on the host cores, fp32, full 4-modality
forward + CrossEntropy + backward + Adam step (the legacy trainer's order of ops,
_archive/AppleCider/core/trainer.py:156-188)."""

from __future__ import annotations

import os
import time

import torch
import torch.nn.functional as F

from . import functional as O


def usable_cores() -> int:
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (a GPU
    box exposes every host core in the affinity mask but grants only a share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    env = os.environ.get("AC_CPU_THREADS")
    if env:
        n = int(env)
    return min(n, 16)  # a 1-GPU box's CPU share


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def time_full_model(state_dict: dict, batch: dict, cfg: dict, steps: int = 3, warmup: int = 1,
                    threads: int | None = None) -> dict:
    """fp32, all usable cores, 1 warm-up + >= 3 timed steps, both
    (i) forward only and (ii) forward + CrossEntropy + backward + Adam step; `value` is (ii)."""
    import sys
    threads = threads or usable_cores()
    torch.set_num_threads(threads)
    params = {k: v.clone().requires_grad_() for k, v in state_dict.items()}
    opt = torch.optim.Adam(list(params.values()), lr=cfg.get("lr", 1e-3))
    args = [torch.from_numpy(batch[k]) for k in ("photometry", "pad_mask", "metadata", "image", "spectra")]
    labels = torch.from_numpy(batch["label"])
    B = labels.shape[0]

    def step():
        opt.zero_grad()
        logits = O.applecider_forward(params, *args, cfg)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        opt.step()
        return float(loss.detach())

    def fwd():
        with torch.no_grad():
            return O.applecider_forward(params, *args, cfg)

    for _ in range(warmup):
        t = time.perf_counter()
        step()
        print(f"[cpu_baseline] warm-up step {time.perf_counter() - t:.1f}s on {threads} threads",
              file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        print(f"[cpu_baseline] timed step {i + 1}/{steps} at {time.perf_counter() - t0:.1f}s",
              file=sys.stderr, flush=True)
    dt = (time.perf_counter() - t0) / steps
    fwd()
    t0 = time.perf_counter()
    for _ in range(steps):
        fwd()
    dtf = (time.perf_counter() - t0) / steps
    return {"value": B / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 fwd+CE+bwd+Adam, batch {B}, {steps} timed steps after {warmup} warm-up",
            "ms_per_step": dt * 1e3, "cpu_model": cpu_model(),
            "fwd_only": {"value": round(B / dtf, 3), "unit": "samples/s", "ms_per_step": round(dtf * 1e3, 1)}}
