import sys, itertools
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from applecider_amd import hipops as H
from applecider_amd.graphstep import GraphedTrainStep
import test_gpu_graphstep as T

dev = torch.device("cuda:0")
H.set_math("bf16x3")
m1, batches = T._fused(dev)
m1.optimizer.prepare().set_capturable(True)
c = H.enable_device_step(dev); c.zero_()
for bt in batches:
    H.step_advance(); T._step_fn(m1, bt)
g_eager = m1.optimizer.fp.grad.clone() if hasattr(m1.optimizer.fp, "grad") else None
p_eager = m1.optimizer.fp.flat.clone()
m2, _ = T._fused(dev)
c.zero_(); m2.optimizer.prepare()
step = GraphedTrainStep(m2, batches[0], step_fn=T._step_fn)
for bt in batches: step(bt)
torch.cuda.synchronize()
d = (m2.optimizer.fp.flat - p_eager).abs()
print("max diff", float(d.max()), "count > 5e-4:", int((d > 5e-4).sum()), "of", d.numel())
fp = m2.optimizer.fp
names = {id(p): n for n, p in m2.named_parameters()}
off = 0
rows = []
for p in m2.parameters():
    n = p.numel()
    st = p.data_ptr() - fp.flat.data_ptr()
    seg = d[st // 4: st // 4 + n]
    if float(seg.max()) > 5e-4:
        gseg = fp.grad[st // 4: st // 4 + n] if hasattr(fp, "grad") else None
        rows.append((float(seg.max()), int((seg > 5e-4).sum()), names[id(p)], tuple(p.shape), float(gseg.abs().max()) if gseg is not None else -1, float(gseg.abs().median()) if gseg is not None else -1))
for r in sorted(rows, reverse=True)[:12]: print(r)
