"""Depthwise 7x7 kernels in isolation (run on the GPU box): time and algorithmic GB/s of the forward
and backward launches at the ConvNeXt stage shapes of the benchmark batch (B = 512)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
VAR = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H._DWCONV_VARIANT = VAR
print(f"variant {VAR} (0 = pipelined LDS-DMA kernels, 1 = round-2 kernels)")
for (S, C) in ((15, 96), (7, 192), (3, 384)):
    x = torch.randn(B, S, S, C, device=dev, requires_grad=True)
    w = torch.randn(49, C, device=dev, requires_grad=True)
    b = torch.randn(C, device=dev, requires_grad=True)
    go = torch.randn(B, S, S, C, device=dev)
    for _ in range(3):
        y = H.dwconv7x7(x, w, b)
        y.backward(go)
    n = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        ev[0].record()
        y = H.dwconv7x7(x, w, b)
        ev[1].record()
        y.backward(go)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    byts = 2.0 * B * S * S * C * 4
    print(f"dwconv {S}x{S}x{C} B={B}: fwd {tf / n * 1e3:7.1f} us = {byts / (tf / n * 1e-3) / 1e9:7.1f} GB/s   "
          f"bwd {tb / n * 1e3:7.1f} us = {1.5 * byts / (tb / n * 1e-3) / 1e9:7.1f} GB/s (dy, x in; dx out)")
