"""A/B of the split-bf16 window-conv kernel variants in ONE process, interleaved rounds (cdna_hip_programming.md rule
24): the SpectraNet conv banks (spectranet.py:18-30) of stages 2-5 at the benchmark batch, forward + input gradient.
    python tools/bench_convx3.py [B=512] [rounds=5] [variants=0,5]
Per (shape, variant): median / min time of the conv_window_x3 launches, algorithmic TFLOP/s."""
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
orig = H.conv_window_x3


def timed(ap, abs_, ars, aco, rb, B_, L, Cw, k, wp, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ok = orig(ap, abs_, ars, aco, rb, B_, L, Cw, k, wp, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw)
    e.record()
    records.setdefault((L, Cw, k, N, bool(flip), H._X3_VARIANT), []).append((s, e, 2.0 * B_ * L * N * k * Cw))
    return ok


H.conv_window_x3 = timed
shapes = ((1024, 64, 128, (3, 31, 251)), (256, 128, 256, (3, 15, 61)), (64, 256, 512, (3, 11, 31)), (16, 512, 1024, (3, 7, 13)))
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
        records.clear()   # warm-up round
H._X3_VARIANT = 0
keys = sorted(set(k[:5] for k in records))
tot = {v: 0.0 for v in variants}
print(f"{'L':>5} {'C':>4} {'k':>4} {'N':>5} {'dir':>4} | " + " | ".join(f"v{v}: med ms   min ms   TF(med)" for v in variants))
for key in keys:
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
    L, C, k, N, flip = key
    print(f"{L:5d} {C:4d} {k:4d} {N:5d} {'dx' if flip else 'fwd':>4} | " + " | ".join(row))
print("sum of medians (ms): " + ", ".join(f"v{v} {tot[v]:.3f}" for v in variants))
