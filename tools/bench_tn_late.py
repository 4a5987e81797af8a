"""Late-stage conv weight gradients (short reductions, wide outputs): tile / split sweep."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import hipops as H
dev = torch.device('cuda')

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

def sweep(B, L, Cin, Cout, k, tiles=(0, 3, 4), splits=(1, 2, 4, 8)):
    P = k // 2; Lp = L + 2 * P
    x16 = H.cast16(torch.randn(B, Lp, Cin, device=dev)); dy16 = H.cast16(torch.randn(B * L, Cout, device=dev))
    dw = torch.zeros(Cout, k * Cin, device=dev)
    out = []
    for tile in tiles:
        for split in splits:
            f = lambda: H.gemm(2, Cout, k * Cin, B * L, H.mat(H._p(dy16), Cout), H.mat(H._p(x16), r1=L, r2=L, s1=Lp * Cin, s3=Cin),
                               H.mat(H._p(dw), k * Cin), accumulate=2, split_k=split, math=2, tile=tile)
            out.append(f"t{tile}s{split}:{timeit(f):.0f}")
    print(f"dW M{Cout} N{k*Cin} K{B*L} plan{H._tn_plan(Cout, k*Cin, B*L)}: " + " ".join(out), flush=True)

sweep(512, 256, 128, 256, 15, (0, 4), (8, 16, 32, 64))
sweep(512, 1024, 64, 128, 31, (0, 4), (8, 16, 32, 64, 128))
sweep(512, 64, 256, 512, 11, (0, 4), (4, 8, 16))
sweep(512, 1024, 64, 128, 251, (0, 4), (4, 8, 16))
sweep(512, 16, 512, 1024, 3, (0, 3, 4), (1, 2, 4))
sweep(512, 64, 256, 512, 3, (0, 3, 4), (2, 4, 8, 16))
sweep(512, 256, 128, 256, 3, (0, 4), (8, 16, 32, 64))
sweep(512, 1024, 64, 128, 3, (0, 4), (32, 64, 128, 256))
