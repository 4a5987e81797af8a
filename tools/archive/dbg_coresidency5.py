"""Whole encoders as the work on the second stream: which branch disturbs the transform kernels, forward or backward?"""
import os, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from applecider_amd import hipops as H
import test_gpu_graphstep as T
dev = torch.device("cuda:0")
H.set_math("bf16x3")
H._MHA_MFMA = not os.environ.get("SCALAR_ATTN")
m, batches = T._fused(dev, B=256)
m.train()
photometry, mask, metadata, images, spectra, labels = batches[0]
g = torch.Generator().manual_seed(0)
x3 = torch.randn(512, 256, 128, generator=g).to(dev)
victim = lambda: H.fft_rows_fwd(x3, None, 0, 256 * 128, 128, 0, 512, 256, 128, 0, (7, 1))
ref = victim()
torch.cuda.synchronize()


def img_fwd():
    with torch.no_grad():
        m.img_metadata_encoder((metadata, images, None))


def img_fwdbwd():
    o = m.img_metadata_encoder((metadata, images, None))
    o.sum().backward()


def pho_fwd():
    with torch.no_grad():
        m.photometry_encoder((photometry, mask, None))


def pho_fwdbwd():
    o = m.photometry_encoder((photometry, mask, None))
    o.sum().backward()


side = torch.cuda.Stream()
for name, f in (("image fwd", img_fwd), ("image fwd+bwd", img_fwdbwd), ("photo fwd", pho_fwd), ("photo fwd+bwd", pho_fwdbwd)):
    bad, worst = 0, 0.0
    for it in range(40):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            f()
        outs = [victim() for _ in range(12)]
        torch.cuda.synchronize()
        e = max(float((o - ref).abs().max()) for o in outs)
        bad += e != 0.0
        worst = max(worst, e)
    print(f"[attention {'scalar' if not H._MHA_MFMA else 'mfma'}] fft_rows 384 beside {name:14s}: {bad:2d} of 40 differ, worst {worst:.3e}", flush=True)
