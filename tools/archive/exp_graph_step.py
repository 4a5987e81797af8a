"""Probe: whole training step through applecider_amd.graphstep.GraphedTrainStep at a given (mode, batch,
encoder streams) — eager vs replay time.  usage: exp_graph_step.py [mode] [B] [streams 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.graphstep import GraphedTrainStep
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
streams = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
dev = torch.device("cuda:0")
H.set_math(mode)
torch.manual_seed(0)
model = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
model.branch_streams = streams
opt = model.optimizer.prepare()
b = make_batch(B, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))

def step():
    return model.train_step(batch)["loss"]

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(3): step()
print(mode, B, streams, "eager ms/step", round(timeit(step), 3), flush=True)
gs = GraphedTrainStep(model, batch, restore_state=False)
print("captured", flush=True)
print(mode, B, streams, "graph ms/step", round(timeit(gs), 3), "loss", float(gs.loss), flush=True)
