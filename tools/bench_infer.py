"""Inference throughput (BASELINE config 5): hipGraph replay vs eager forward, alerts/s.
    python tools/bench_infer.py [batch=2048] [math=bf16]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.inference import GraphedClassifier
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
math_mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
H.set_math(math_mode)
torch.manual_seed(0)
net = AppleCider(dict(bench.FUSION_CFG)).to(dev)
net.optimizer.prepare()
gc = GraphedClassifier(net, batch_size=B, use_probabilities=True)
b = make_batch(B, seed=4)
batch = {k: torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra")}

def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

n = 10 if B >= 1024 else 30
tg = timeit(lambda: gc.predict(batch), n)
te = timeit(lambda: gc.eager(batch), n)
same = torch.equal(gc.predict(batch).clone(), gc.eager(batch))
print(json.dumps({"workload": "BASELINE configs[4]: inference, full 4-modality forward + softmax, inputs resident in HBM",
                  "batch": B, "mfma_input_dtype": math_mode, "graph_ms": round(tg * 1e3, 3),
                  "eager_ms": round(te * 1e3, 3), "alerts_per_s_graph": round(B / tg, 1),
                  "alerts_per_s_eager": round(B / te, 1), "graph_equals_eager": bool(same)}))
