"""configs[1] (AstroMiNN, B = 256, eager step) with module switches flipped, interleaved in one process: which part of the
step a regression of that leg comes from.  usage: ab_astro.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.config import default_config
from applecider_amd.models.astrominn import AstroMiNN
from applecider_amd.synthetic import make_batch
dev = torch.device("cuda:0")
H.set_math("bf16x3")
torch.manual_seed(1)
net = AstroMiNN(default_config()).to(dev).train()
b = make_batch(256, seed=1)
bt = tuple(torch.from_numpy(b[k]).to(dev) for k in ("metadata", "image", "target"))
opt = net.this_optimizer.prepare()

def step():
    opt.zero_grad()
    loss = net.this_criterion(net(bt), bt[2])
    loss.backward()
    opt.step()

def timeit(n=20):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def gpu_time(n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): step()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for rnd in range(3):
    for tag, pool in (("zero pool on ", True), ("zero pool off", False)):
        H._ZERO_POOL = pool
        print(f"round {rnd} {tag}: wall {timeit():.3f} ms/step  (events {gpu_time():.3f})", flush=True)
H._ZERO_POOL = True
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
tot = sum(ev.device_time for ev in prof.key_averages()) / 3 / 1e3
print(f"device kernel time per step {tot:.3f} ms over {sum(ev.count for ev in prof.key_averages()) / 3:.0f} launches")
