"""Probe: spectra branch and the two small branches on disjoint CU sets (hipExtStreamCreateWithCUMask).
The big kernels of the spectra branch hold every CU they run on (160 KB LDS, 256 VGPRs x 8 waves), so the small
branches on plain streams only time-slice with them; on their own CUs they run truly concurrently."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
H.set_math(mode)
torch.manual_seed(0)
m = AppleCider(dict(bench.FUSION_CFG)).to(dev).train()
m.optimizer.prepare()
b = make_batch(512, seed=2)
batch = tuple(torch.from_numpy(b[k]).to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))
hip = ctypes.CDLL("libamdhip64.so")

def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)

def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def step():
    return m.train_step(batch)["loss"]

print(mode, "plain three streams", round(timeit(step), 2), "ms", flush=True)
def words(fn):
    out = []
    for w in range(8):
        v = 0
        for bit in range(32):
            if fn(32 * w + bit): v |= 1 << bit
        out.append(v)
    return out
parts = {
    "3 of every 32 bits": (lambda i: i % 32 < 29, lambda i: i % 32 >= 29),
    "4 of every 32 bits": (lambda i: i % 32 < 28, lambda i: i % 32 >= 28),
    "6 of every 32 bits": (lambda i: i % 32 < 26, lambda i: i % 32 >= 26),
    "last 24 bits": (lambda i: i < 232, lambda i: i >= 232),
    "last 32 bits": (lambda i: i < 224, lambda i: i >= 224),
    "bits = 7 mod 8 (32)": (lambda i: i % 8 != 7, lambda i: i % 8 == 7),
}
for name, (big, small) in parts.items():
    try:
        s_big = masked_stream(words(big))
        s_small = [masked_stream(words(small)) for _ in range(2)]
        m._branch_streams = s_small
        H._side_streams.clear()
        H.register_side_streams(s_small)
        def f():
            with torch.cuda.stream(s_big):
                loss = step()
            return loss
        ms = timeit(f)
        torch.cuda.synchronize()
        print(mode, name, round(ms, 2), "ms", flush=True)
    except Exception as ex:
        print(name, "failed:", type(ex).__name__, ex, flush=True)
