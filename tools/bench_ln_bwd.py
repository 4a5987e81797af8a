"""LayerNorm backward at the step's two hot row shapes (photometry encoder 66 048 x 128, ConvNeXt stage 0 115 200 x 96),
time per launch.  usage: [APPLECIDER_HIP_LIB=other.so] bench_ln_bwd.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')
H.set_math("bf16x3")
for rows, C in ((66048, 128), (115200, 96), (25088, 192), (131072, 768)):
    x = torch.randn(rows, C, device=dev, requires_grad=True)
    g = torch.nn.Parameter(torch.ones(C, device=dev)); b = torch.nn.Parameter(torch.zeros(C, device=dev))
    go = torch.randn(rows, C, device=dev)
    y = H.layer_norm(x, g, b, 1e-6)
    for _ in range(3):
        y.backward(go, retain_graph=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    s.record()
    for _ in range(n):
        y.backward(go, retain_graph=True)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    print(f"{os.environ.get('APPLECIDER_HIP_LIB', 'default'):>28}  rows {rows:7d} C {C:4d}: {us:7.1f} us  {3 * rows * C * 4 / us / 1e6:6.2f} TB/s", flush=True)
