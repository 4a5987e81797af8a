"""Per-shape time of the matrix-core launches in one training step (run on the GPU box).

Wraps hipops.gemm / hipops.conv_window with HIP events and prints, per (mode, M, N, K), the number of
launches, summed time per step and the achieved TFLOP/s — the list that says which shapes to tune.
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench

dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H.set_math(sys.argv[2] if len(sys.argv) > 2 else "bf16")
H._X3_VARIANT = int(os.environ.get("APPLECIDER_X3_VARIANT", "0"))
torch.manual_seed(0)
net = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
net.branch_streams = False   # one stream: a launch's HIP-event bracket then times that launch alone
net.optimizer.prepare()
b = make_batch(B, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in
              ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))

recs = {}
on = [False]
og, ow, owg = H.gemm, H.conv_window, H.conv_wgrad


def twg(dy, dy_lo, dbs, drs, drb, dco, x, x_lo, xbs, xrs, xrb, xrows, Bn, L, Cout, Cin, k, dw, **kw):
    if not on[0]:
        return owg(dy, dy_lo, dbs, drs, drb, dco, x, x_lo, xbs, xrs, xrb, xrows, Bn, L, Cout, Cin, k, dw, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ok = owg(dy, dy_lo, dbs, drs, drb, dco, x, x_lo, xbs, xrs, xrb, xrows, Bn, L, Cout, Cin, k, dw, **kw); e.record()
    if ok:
        recs.setdefault(("WGR", Cout, k * Cin, Bn * L, "-", 1), []).append((s, e, 2.0 * Bn * L * Cout * k * Cin))
    return ok



def tg(mode, M, N, K, a, b, c, **kw):
    if not on[0]:
        return og(mode, M, N, K, a, b, c, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); og(mode, M, N, K, a, b, c, **kw); e.record()
    key = ("NT NN TN".split()[mode], M, N, K, "g" if (a.rows.r1 or b.rows.r1 or a.goff or b.goff) else "-",
           kw.get("split_k", 1))
    recs.setdefault(key, []).append((s, e, 2.0 * M * N * K))


def tw(a16, abs_, ars, aco, rb, Bn, L, Cw, k, w16, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw):
    if not on[0]:
        return ow(a16, abs_, ars, aco, rb, Bn, L, Cw, k, w16, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ok = ow(a16, abs_, ars, aco, rb, Bn, L, Cw, k, w16, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw); e.record()
    if ok:
        recs.setdefault(("WIN", Bn * L, N, k * Cw, "f" if flip else "-", 1), []).append((s, e, 2.0 * Bn * L * N * k * Cw))
    return ok


owx = H.conv_window_x3
sup = [False]


def twx(ap, abs_, ars, aco, rb, Bn, L, Cw, k, wp, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw):
    if not on[0]:
        return owx(ap, abs_, ars, aco, rb, Bn, L, Cw, k, wp, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    on[0] = False   # inner conv_window launches (3-pass fallback) are part of this record
    s.record(); ok = owx(ap, abs_, ars, aco, rb, Bn, L, Cw, k, wp, wrs, wts, flip, N, c_ptr, ldc, bias, acc, **kw); e.record()
    on[0] = True
    if ok:
        recs.setdefault(("WX3", Bn * L, N, k * Cw, "t" if kw.get("tap_row_step") else ("f" if flip else "-"), 1), []).append((s, e, 2.0 * Bn * L * N * k * Cw))
    return ok


H.gemm, H.conv_window, H.conv_wgrad, H.conv_window_x3 = tg, tw, twg, twx
for i in range(3):
    on[0] = i == 2
    loss = net.train_step(batch)["loss"]
torch.cuda.synchronize()
rows = []
for key, r in recs.items():
    ms = sum(s.elapsed_time(e) for s, e, _ in r)
    fl = sum(w for _, _, w in r)
    rows.append((ms, key, len(r), fl))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total matrix-core launch time {tot:.2f} ms over {sum(r[2] for r in rows)} launches")
for ms, key, n, fl in rows[:60]:
    print(f"{ms:8.3f} ms  n={n:3d}  {key[0]:3s} M={key[1]:7d} N={key[2]:6d} K={key[3]:7d} {key[4]} split={key[5]}  {fl / ms / 1e9:7.1f} TF")
