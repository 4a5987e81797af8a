"""Where do the small ATen launches of one training step come from (add / copy / fill / zero / clone)?
torch.profiler over one step after warm-up, CPU ops grouped by name and by the innermost applecider_amd frame of their
Python stack (ops issued by the autograd engine itself have no Python frame: they are listed under their parent node)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from torch.profiler import profile, ProfilerActivity
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench

dev = torch.device('cuda')
H.set_math("bf16x3")
torch.manual_seed(0)
net = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
net.branch_streams = False
net.optimizer.prepare()
b = make_batch(512, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))
for _ in range(3):
    net.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    net.train_step(batch)
    torch.cuda.synchronize()
want = ("aten::add", "aten::add_", "aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like",
        "aten::contiguous", "aten::cat", "aten::sum", "aten::mul", "aten::to", "aten::_to_copy", "aten::empty_like")
by = collections.Counter()
evs = prof.events()
for e in evs:
    if e.name not in want:
        continue
    if e.cpu_parent is not None and e.cpu_parent.name in want:
        continue   # count the outermost ATen op only
    site = None
    for fr in (e.stack or []):
        if "applecider_amd" in fr and "hipops.py" not in fr.split("(")[0][-10:]:
            site = fr
            break
    if site is None:
        for fr in (e.stack or []):
            if "applecider_amd" in fr:
                site = fr
                break
    if site is None:
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name)
            par = par.cpu_parent
        site = "engine: " + " < ".join(chain)
    ndev = len(e.kernels) if hasattr(e, "kernels") else 0
    by[(e.name, site.strip()[-110:], ndev > 0)] += 1
print("count  op  launches-a-kernel  site")
for (name, site, k), n in sorted(by.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n:5d}  {name:18s} {str(k):5s}  {site}")
kern = collections.Counter()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA and ("elementwise" in e.name or "copyBuffer" in e.name or "fillBuffer" in e.name or "Memcpy" in e.name or "Memset" in e.name):
        kern[e.name[:90]] += 1
print("\ndevice-side small launches of this step:")
for k, n in kern.most_common(20):
    print(f"{n:5d}  {k}")
