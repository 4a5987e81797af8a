"""Per-call-site GPU time of one training step (run on the GPU box).

Every C-ABI entry point is wrapped with HIP events and attributed to the Python call site
(function:line two frames up), so the table shows WHICH cast / LayerNorm / column-sum costs what.
"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H, _lib
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench

dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H.set_math("bf16")
torch.manual_seed(0)
net = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
net.branch_streams = False    # one stream: per-launch times without overlap
net.optimizer.prepare()
b = make_batch(B, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in
              ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))

real = _lib.load()
recs = collections.defaultdict(list)
on = [False]


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("ac_"):
            return fn

        def timed(*a):
            if not on[0]:
                return fn(*a)
            f = sys._getframe(1)
            site = f"{f.f_code.co_name}:{f.f_lineno}"
            f2 = f.f_back
            if f2 is not None and f.f_code.co_name in ("gemm", "cast16", "cast16_T", "cast16_w", "cast16_wT", "colsum", "conv_window", "_weight_grad", "_bias_grad"):
                site += f" <- {f2.f_code.co_name}:{f2.f_lineno}"
                f3 = f2.f_back
                if f3 is not None and f2.f_code.co_name in ("cast16_w", "cast16_wT", "_weight_grad", "_bias_grad", "timed"):
                    site += f" <- {f3.f_code.co_name}:{f3.f_lineno}"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rc = fn(*a); e.record()
            recs[(name, site)].append((s, e))
            return rc
        return timed


proxy = Proxy()
H._lib_ = lambda: proxy
for i in range(3):
    on[0] = i == 2
    loss = net.train_step(batch)["loss"]
torch.cuda.synchronize()
rows = sorted(((sum(s.elapsed_time(e) for s, e in v), k, len(v)) for k, v in recs.items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"total {tot:.2f} ms over {sum(r[2] for r in rows)} launches")
by = collections.Counter()
for ms, (name, site), n in rows:
    by[name] += ms
print("by entry point:", ", ".join(f"{k} {v:.2f}" for k, v in by.most_common(25)))
for ms, (name, site), n in rows[:90]:
    print(f"{ms:7.3f} ms n={n:3d} {name:24s} {site}")
