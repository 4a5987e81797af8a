import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from applecider_amd import hipops as H
import test_gpu_graphstep as T

dev = torch.device("cuda:0")
H.set_math("bf16x3")
import os
orig = H.fftconv_covered
if os.environ.get("ONLYK"):
    H.fftconv_covered = lambda B, L, Cin, Cout, k: k == int(os.environ["ONLYK"]) and orig(B, L, Cin, Cout, k)
print("plan k=251", H.fft_plan(1024, 251), "only k", os.environ.get("ONLYK"))
m, batches = T._fused(dev, B=int(os.environ.get('BATCH', '8')))
m.eval()
bt = batches[0]


def fwd():
    with torch.no_grad():
        return torch.cat([t.clone() for t in m.get_embeddings(*bt[:5])], 1)


m.branch_streams = False
a = fwd(); a2 = fwd()
m.branch_streams = True
b = fwd(); b2 = fwd()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd(); fwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
res = []
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    res.append(out.clone())
d = lambda x, y: float((x - y).abs().max())
for nm, sl in (("photo", slice(0, 64)), ("image", slice(64, 128)), ("spectra", slice(128, 192))):
    print(nm, "graph vs 1s", [d(r[:, sl], a[:, sl]) for r in res], "3s vs 1s", d(b[:, sl], a[:, sl]))
print("1s vs 1s", d(a, a2), "3s vs 1s", d(b, a), "3s vs 3s", d(b, b2), "graph vs 1s", [d(r, a) for r in res], "scale", float(a.abs().max()))
H._FFTCONV = False
a0 = fwd()
print("fft vs direct (1 stream... 3s)", d(a0, a))
