"""cProfile of the host side of a training step at a tiny batch (GPU work negligible)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench
dev = torch.device('cuda')
H.set_math("bf16")
net = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
net.optimizer.prepare()
b = make_batch(16, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))
for _ in range(3):
    net.train_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    net.train_step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
