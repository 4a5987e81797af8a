"""Driver for rocprofv3 counter passes over the split-bf16 conv kernels: the SpectraNet
stage-2 and stage-3 conv banks (spectranet.py:18-30; default_config.toml:104-114) at the benchmark batch, forward +
backward, so that conv1d_window_x3_kernel (forward, input gradient) and conv1d_wgrad_kernel<SPLIT> run at the shapes
that hold 26 of the step's 55 ms.  Nothing else runs: PMC passes serialise every dispatch.
    python tools/pmc_conv.py [B=512] [reps=3] [variant=0]"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from applecider_amd import hipops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H._X3_VARIANT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
H.set_math("bf16x3")
torch.manual_seed(0)
for (L, Cin, Cout, ks) in ((1024, 64, 128, (3, 31, 251)), (256, 128, 256, (3, 15, 61))):
    x = torch.randn(B, L, Cin, device=dev).requires_grad_()
    ws = [torch.nn.Parameter(torch.randn(Cout, k * Cin, device=dev) / math.sqrt(Cin * k)) for k in ks]
    bs = [torch.nn.Parameter(torch.randn(Cout, device=dev)) for _ in ks]
    go = torch.randn(B, L, 3 * Cout, device=dev)
    for _ in range(reps):
        y = H.conv_group1d(x, ks, ws, bs)
        y.backward(go)
        x.grad = None
    torch.cuda.synchronize()
    del x, ws, bs, go, y
print("done")
