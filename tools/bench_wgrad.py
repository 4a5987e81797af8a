"""A/B of the split-bf16 weight-gradient kernels in one process, interleaved rounds: SpectraNet conv banks of stages
2-4 at the benchmark batch (spectranet.py:18-20; torch's conv1d backward).
    python tools/bench_wgrad.py [B=512] [rounds=5] [variants=4,0]"""
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from applecider_amd import hipops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "4,0").split(",")]
dev = torch.device("cuda:0")
H.set_math("bf16x3")
torch.manual_seed(0)
records = {}
orig = H.conv_wgrad


def timed(dy, dy_lo, dbs, drs, drb, dco, x, x_lo, xbs, xrs, xrb, xrows, B_, L, Cout, Cin, k, dw, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ok = orig(dy, dy_lo, dbs, drs, drb, dco, x, x_lo, xbs, xrs, xrb, xrows, B_, L, Cout, Cin, k, dw, **kw)
    e.record()
    if ok:
        records.setdefault((L, Cin, Cout, k, H._X3_VARIANT), []).append((s, e, 2.0 * B_ * L * Cout * k * Cin))
    return ok


H.conv_wgrad = timed
shapes = ((1024, 64, 128, (3, 31, 251)), (256, 128, 256, (3, 15, 61)), (64, 256, 512, (3, 11, 31)))
cases = []
for (L, Cin, Cout, ks) in shapes:
    x = torch.randn(B, L, Cin, device=dev).requires_grad_()
    ws = [torch.nn.Parameter(torch.randn(Cout, k * Cin, device=dev) / math.sqrt(Cin * k)) for k in ks]
    bs = [torch.nn.Parameter(torch.randn(Cout, device=dev)) for _ in ks]
    go = torch.randn(B, L, 3 * Cout, device=dev)
    cases.append((x, ws, bs, go, ks))
for r in range(rounds + 1):
    for v in variants:
        H._X3_VARIANT = v
        for (x, ws, bs, go, ks) in cases:
            y = H.conv_group1d(x, ks, ws, bs)
            y.backward(go)
            x.grad = None
        torch.cuda.synchronize()
    if r == 0:
        records.clear()
H._X3_VARIANT = 0
tot = {v: 0.0 for v in variants}
print(f"{'L':>5} {'Cin':>4} {'Cout':>5} {'k':>4} | " + " | ".join(f"v{v}: med ms   min ms   TF(med)" for v in variants))
for key in sorted(set(k[:4] for k in records)):
    row = []
    for v in variants:
        recs = records.get(key + (v,), [])
        ms = [s.elapsed_time(e) for s, e, _ in recs]
        if not ms:
            row.append("      -        -        -")
            continue
        med = statistics.median(ms)
        tot[v] += med
        row.append(f"   {med:7.4f}  {min(ms):7.4f}  {recs[0][2] / (med * 1e-3) / 1e12:7.1f}")
    print(f"{key[0]:5d} {key[1]:4d} {key[2]:5d} {key[3]:4d} | " + " | ".join(row))
print("sum of medians (ms): " + ", ".join(f"v{v} {tot[v]:.3f}" for v in variants))
