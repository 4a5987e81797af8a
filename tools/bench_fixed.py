"""Fixed cost of one TN launch: accumulate mode x grid size at K = 512 (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

def run(mode, M, N, K, accs, split=1):
    if mode == "TN":
        a16, b16 = H.cast16(torch.randn(K, M, device=dev)), H.cast16(torch.randn(K, N, device=dev))
        am, bm = H.mat(H._p(a16), M), H.mat(H._p(b16), N)
        md = H.AC_GEMM_TN
    else:
        a16, b16 = H.cast16(torch.randn(M, K, device=dev)), H.cast16(torch.randn(N, K, device=dev))
        am, bm = H.mat(H._p(a16), K), H.mat(H._p(b16), K)
        md = H.AC_GEMM_NT
    c = torch.zeros(M, N, device=dev)
    out = []
    for acc in accs:
        f = lambda: H.gemm(md, M, N, K, am, bm, H.mat(H._p(c), N), math=2, accumulate=acc, split_k=split)
        out.append(f"acc{acc}:{timeit(f):.1f}us")
    print(f"{mode} M{M} N{N} K{K} split{split} ({-(-M//128) * -(-N//128) * split} WGs): " + "  ".join(out), flush=True)

empty = torch.zeros(1, device=dev)
print("empty-kernel floor (ac_add n=1):", timeit(lambda: H._lib_().ac_add(H._p(empty), H._p(empty), H._p(empty), 1, 1.0, H._stream())), "us")
for mode in ("TN", "NT"):
    for (M, N) in ((128, 128), (256, 256), (768, 768), (768, 3072), (2048, 2048), (2048, 4096)):
        run(mode, M, N, 512, [0, 1, 2])
    run(mode, 768, 3072, 4096, [0, 1, 2])
run("TN", 384, 1536, 4608, [0, 2], split=1)
run("TN", 384, 1536, 4608, [2], split=9)
run("TN", 128, 128, 66048, [2], split=64)
run("TN", 128, 128, 66048, [2], split=8)
