"""Probe: forward + backward time of each encoder of the fused model on its own (B = 512, one stream) — how much
of the step is the spectra branch and what the overlap of the three streams recovers.
usage: exp_branch_times.py [mode]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0")
H.set_math(mode)
torch.manual_seed(0)
m = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
m.branch_streams = False
opt = m.optimizer.prepare()
b = make_batch(512, seed=2)
t = {k: torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label")}

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def run(enc, *args):
    def f():
        opt.zero_grad()
        out = enc(*args)
        out.sum().backward()
    return f

print(mode, "spectra  ", round(timeit(run(m.spectra_encoder, (t["spectra"], None, None))), 2), "ms", flush=True)
print(mode, "image+md ", round(timeit(run(m.img_metadata_encoder, (t["metadata"], t["image"], None))), 2), "ms", flush=True)
print(mode, "photo    ", round(timeit(run(m.photometry_encoder, (t["photometry"], t["pad_mask"], None))), 2), "ms", flush=True)
def full():
    m.train_step(tuple(t[k] for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label")))
print(mode, "full step, one stream   ", round(timeit(full), 2), "ms", flush=True)
m.branch_streams = True
print(mode, "full step, three streams", round(timeit(full), 2), "ms", flush=True)
# A/B of the stream layout (round 4): both side branches on ONE side stream; side streams at high priority
import applecider_amd.models.applecider as AC
st = m._streams(dev)
orig = list(st)
m._branch_streams = [st[0], st[0]]
print(mode, "full step, two streams (image + photometry share one)", round(timeit(full), 2), "ms", flush=True)
lo, hi = torch.cuda.Stream.priority_range()
m._branch_streams = [torch.cuda.Stream(device=dev, priority=hi), torch.cuda.Stream(device=dev, priority=hi)]
H.register_side_streams(m._branch_streams)
print(mode, "full step, three streams, side branches at high priority", round(timeit(full), 2), "ms", flush=True)
m._branch_streams = orig
print(mode, "full step, three streams (again)", round(timeit(full), 2), "ms", flush=True)
