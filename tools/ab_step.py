"""A/B of module-level hipops switches on the full B = 512 step, same process, same box.
usage: python tools/ab_step.py _FUSE_DACT [_OTHER ...]    (each switch: step time with True, then False)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench
dev = torch.device("cuda:0")
H.set_math("bf16x3")
torch.manual_seed(0)
m = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
m.optimizer.prepare()
b = make_batch(512, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))

def timeit(n=10):
    for _ in range(3): m.train_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.train_step(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for name in sys.argv[1:]:
    for rep in range(2):
        for val in (True, False):
            setattr(H, name, val)
            H.clear_step_cache()
            print(f"{name} = {val}: {timeit():.2f} ms/step (three streams)", flush=True)
    setattr(H, name, True)
