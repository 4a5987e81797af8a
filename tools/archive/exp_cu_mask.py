"""Probe: does hipExtStreamCreateWithCUMask confine a stream's kernels on this device, and how does a large
product scale with the number of CUs it may use?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")

def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)

H.set_math("bf16x3")
M, N, K = 262144, 512, 1028
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
def work():
    H.gemm(H.AC_GEMM_NT, M, N, K, H.mat(H._p(a), K), H.mat(H._p(b), K), H.mat(H._p(c), N), math=3)
def timeit(stream, n=5):
    with torch.cuda.stream(stream):
        work(); stream.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(n): work()
        e.record(stream); stream.synchronize()
    return s.elapsed_time(e) / n
print("default stream", round(timeit(torch.cuda.current_stream()), 3), "ms", flush=True)
for name, words in [("all 256", [0xFFFFFFFF] * 8), ("first 128 bits", [0xFFFFFFFF] * 4 + [0] * 4),
                    ("every other bit (128)", [0x55555555] * 8), ("low 29 bits of each word (232)", [0x1FFFFFFF] * 8),
                    ("high 3 bits of each word (24)", [0xE0000000] * 8), ("first 32 bits", [0xFFFFFFFF] + [0] * 7)]:
    try:
        st = masked_stream(words)
        print(name, round(timeit(st), 3), "ms", flush=True)
    except Exception as ex:
        print(name, "failed:", ex, flush=True)
