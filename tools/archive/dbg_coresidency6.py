"""Co-residency question: transform workgroups with their EXACT LDS request (APPLECIDER_FFT_SHARED_CU=1,
small-LDS shapes: channel counts that are no multiple of 64 take 8 sequences per 512-thread workgroup) beside the
matrix-core attention kernel on a second stream, 60 runs each.  Run once per library build (TAG names it):
the product library, and the diagnostic builds of ac_fft.hip with two barriers in a row / barrier + sleep."""
import os, sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H
dev = torch.device("cuda:0")
H.set_math("bf16x3")
g = torch.Generator().manual_seed(0)
x3 = torch.randn(512, 256, 112, generator=g).to(dev)
xl = torch.randn(64, 1024, 48, generator=g).to(dev)
qkv = torch.randn(512, 129, 384, device=dev); pad = torch.zeros(512, 129, dtype=torch.uint8, device=dev)
victims = {"fft_rows 384 (C=112)": lambda: H.fft_rows_fwd(x3, None, 0, 256 * 112, 112, 0, 512, 256, 112, 0, (7, 1)),
           "fft_rows 512 (C=112)": lambda: H.fft_rows_fwd(x3, None, 0, 256 * 112, 112, 0, 512, 256, 112, 0, 9),
           "fft_rows 288 (C=112)": lambda: H.fft_rows_fwd(x3, None, 0, 256 * 112, 112, 0, 512, 256, 112, 0, (5, 2)),
           "fft_rows 1536 (C=48)": lambda: H.fft_rows_fwd(xl, None, 0, 1024 * 48, 48, 0, 64, 1024, 48, 0, (9, 1))}
side = torch.cuda.Stream()
for vn, vf in victims.items():
    ref = vf()
    torch.cuda.synchronize()
    bad, worst, nbad = 0, 0.0, 0
    for it in range(60):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.no_grad():
                for _ in range(4):
                    H.mha(qkv, pad, 8, 0.0, False)
        out = vf()
        torch.cuda.synchronize()
        d = (out != ref)
        e = float((out - ref).abs().max())
        bad += e != 0.0
        nbad += int(d.sum())
        worst = max(worst, e)
    print(f"[{os.environ.get('TAG', 'product')}] {vn:22s} beside attention: {bad:3d} of 60 runs differ, {nbad} values in all, worst {worst:.3e}", flush=True)
