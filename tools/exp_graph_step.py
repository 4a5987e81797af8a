"""Experiment: upper bound of capturing the whole training step in a HIP graph (seeds and the Adam step
count are baked in, so this is a timing probe only, not a training mode)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench
dev = torch.device("cuda:0")
H.set_math("bf16")
torch.manual_seed(0)
model = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
opt = model.optimizer.prepare()
b = make_batch(512, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))

def step():
    opt.zero_grad()
    loss = H.cross_entropy_index(model(*batch[:5]), batch[5])
    loss.backward()
    opt.step()
    return loss

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(3): step()
print("eager ms/step", round(timeit(step), 3), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
print("graph ms/step", round(timeit(g.replay), 3), "loss", float(static_loss), flush=True)
