"""Which tensor of the frequency-domain forward chain first differs inside the three-branch hipGraph?"""
import sys, os
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from applecider_amd import hipops as H
import test_gpu_graphstep as T

dev = torch.device("cuda:0")
H.set_math("bf16x3")
orig = H.fftconv_covered
K = int(os.environ.get("ONLYK", "251"))
H.fftconv_covered = lambda B, L, Cin, Cout, k: k == K and orig(B, L, Cin, Cout, k)
m, batches = T._fused(dev)
m.eval()
bt = batches[0]
rec = []
o_rf, o_tf, o_gb, o_ri = H.fft_rows_fwd, H.fft_taps_fwd, H.gemm_batched, H.fft_rows_inv


def rf(*a, **k):
    r = o_rf(*a, **k); rec.append(("rows_fwd", r.clone())); return r


_static = {}


def tf(w, Cout, Cin, k, size):
    if os.environ.get("STATIC_HB"):
        # the same launch into a persistent buffer (outside the graph's pool)
        logm, r3, N = H._fft_size(size)
        key = (Cout, Cin, k, N)
        if key not in _static:
            _static[key] = torch.empty(N // 2 + 1, 2 * Cout, 2 * Cin, device=w.device)
        hb = _static[key]
        from applecider_amd import _lib
        _lib.check(H._lib_().ac_fft_taps_fwd(H._p(w), Cout, Cin, k, logm, r3, H._p(H._fft_tw(size, w.device)), H._p(hb), H._stream()), "taps")
        r = hb
    else:
        r = o_tf(w, Cout, Cin, k, size)
    rec.append(("taps_fwd", r.clone())); rec.append(("w", w.clone())); return r


def ri(spec, B, Cn, size, dst, *a, **k):
    rec.append(("prod(yf)", spec.clone()))
    o_ri(spec, B, Cn, size, dst, *a, **k)
    rec.append(("rows_inv(ycat)", dst.clone()))


H.fft_rows_fwd, H.fft_taps_fwd, H.fft_rows_inv = rf, tf, ri


def fwd():
    rec.clear()
    with torch.no_grad():
        out = torch.cat([t.clone() for t in m.get_embeddings(*bt[:5])], 1)
    return out, list(rec)


m.branch_streams = True
ref, ref_rec = fwd()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out, grec = fwd()
for it in range(3):
    g.replay(); torch.cuda.synchronize()
    print("replay", it, "final", float((out - ref).abs().max()),
          [(n, float((a - b).abs().max()), float(b.abs().max())) for (n, a), (_, b) in zip(grec, ref_rec)], flush=True)
