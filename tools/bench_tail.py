"""Times the fused SpectraNet tail kernels (csrc/ac_tail.hip) at the shapes of the B = 512 step and rates them against
their algorithmic bytes.  usage: python tools/bench_tail.py [reps]"""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from applecider_amd import _lib, hipops as H

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
H.set_math("bf16x3")
lib = _lib.load()
for (rows, K, N, L, pad) in [(512 * 4096, 192, 64, 4096, 0), (512 * 1024, 384, 128, 1024, 125)]:
    g = torch.Generator().manual_seed(0)
    ycat = (torch.randn(rows // 64, K, generator=g).repeat(64, 1) * 2 + 0.3).to(dev)
    gam, bet = (1 + 0.2 * torch.randn(K, generator=g)).to(dev), (0.1 * torch.randn(K, generator=g)).to(dev)
    w, b = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev), torch.randn(N, generator=g).to(dev)
    dpool = torch.randn(rows // 4, N, device=dev)
    wh, wl = H.split16(w)
    wth, wtl = H.split16(w.t().contiguous())
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    pooled = torch.empty(rows // 4, N, device=dev)
    idx = torch.empty(rows // 4, N, device=dev, dtype=torch.uint8)
    Lp = L + 2 * pad
    planes = torch.zeros(2, rows // L, Lp, K, device=dev, dtype=torch.bfloat16)
    dgam, dbet, dxs, dw = torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.zeros(N, K, device=dev)
    P = H._p
    st = H._stream()
    fns = {
        "fwd": (lambda: lib.ac_spectail_fwd(P(ycat), P(gam), P(bet), 1e-5, P(wh), P(wl), P(b), P(mean), P(rstd), P(pooled), P(idx), rows, K, N, st),
                rows * K * 4 + rows // 4 * N * 5 + rows * 8),
        "bwd_dx": (lambda: lib.ac_spectail_bwd_dx(P(ycat), P(mean), P(rstd), P(gam), P(bet), P(dpool), P(idx), P(wth), P(wtl), P(planes[0]), P(planes[1]),
                                                  L if pad else 0, Lp if pad else 0, pad, P(dgam), P(dbet), P(dxs), rows, K, N, st),
                   rows * K * 8 + rows // 4 * N * 5 + rows * 8),
        "bwd_dw": (lambda: lib.ac_spectail_bwd_dw(P(ycat), P(mean), P(rstd), P(gam), P(bet), P(dpool), P(idx), P(dw), rows, K, N, st),
                   rows * K * 4 + rows // 4 * N * 5 + rows * 8),
    }
    for name, (fn, nbytes) in fns.items():
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        print(f"{name:7s} rows={rows} K={K} N={N}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:7.1f} GB/s (algorithmic {nbytes / 1e9:.2f} GB)", flush=True)
