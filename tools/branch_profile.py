"""Run forward + backward of ONE encoder of the fused model (B = 512, one stream) a few times, for
`rocprofv3 --kernel-trace --stats -- python3 tools/branch_profile.py spectra 5`: the kernel statistics of the branch
that sets the step's critical path.  usage: branch_profile.py {spectra|image|photo} [steps] [mode]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "spectra"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
dev = torch.device("cuda:0")
H.set_math(mode)
torch.manual_seed(0)
m = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
m.branch_streams = False
opt = m.optimizer.prepare()
b = make_batch(512, seed=2)
t = {k: torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label")}
enc, args = {"spectra": (m.spectra_encoder, ((t["spectra"], None, None),)),
             "image": (m.img_metadata_encoder, ((t["metadata"], t["image"], None),)),
             "photo": (m.photometry_encoder, ((t["photometry"], t["pad_mask"], None),))}[which]
for _ in range(steps):
    opt.zero_grad()
    enc(*args).sum().backward()
torch.cuda.synchronize()
print(which, steps, "steps done")
