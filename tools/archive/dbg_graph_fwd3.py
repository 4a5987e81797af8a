"""Which side branch, running beside the spectra encoder inside a hipGraph, disturbs the frequency-domain kernels?"""
import sys, os
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from applecider_amd import hipops as H
import test_gpu_graphstep as T

dev = torch.device("cuda:0")
H.set_math("bf16x3")
m, batches = T._fused(dev)
m.eval()
photometry, mask, metadata, images, spectra, labels = batches[0]
side = [torch.cuda.Stream() for _ in range(2)]


def run(which):
    main = torch.cuda.current_stream()
    outs = []
    with torch.no_grad():
        for st, name in zip(side, ("image", "photo")):
            if name in which:
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    if name == "image":
                        outs.append(m.img_metadata_encoder((metadata, images, None)))
                    else:
                        outs.append(m.photometry_encoder((photometry, mask, None)))
        s = m.spectra_encoder((spectra, None, None)).clone()
        for st, name in zip(side, ("image", "photo")):
            if name in which:
                main.wait_stream(st)
    return s, outs


ref, _ = run(())
torch.cuda.synchronize()
for which in ((), ("image",), ("photo",), ("image", "photo")):
    w = torch.cuda.Stream(); w.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(w):
        run(which)
    torch.cuda.current_stream().wait_stream(w); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, keep = run(which)
    errs = []
    for _ in range(4):
        g.replay(); torch.cuda.synchronize()
        errs.append(float((out - ref).abs().max()))
    print("beside", which or "nothing", errs, flush=True)
