import sys, itertools
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from applecider_amd import hipops as H
from applecider_amd.graphstep import GraphedTrainStep
import test_gpu_graphstep as T

dev = torch.device("cuda:0")


def scenario(tag, streams=True, ln_planes=True, fft=True):
    H.set_math("bf16x3")
    H._FFTCONV, H._LN_PLANES = fft, ln_planes
    m1, batches = T._fused(dev)
    m1.branch_streams = streams
    m1.optimizer.prepare().set_capturable(True)
    c = H.enable_device_step(dev)
    c.zero_()
    eager = []
    for bt in batches:
        H.step_advance()
        eager.append(float(T._step_fn(m1, bt)))
    m2, _ = T._fused(dev)
    m2.branch_streams = streams
    c.zero_()
    m2.optimizer.prepare()
    step = GraphedTrainStep(m2, batches[0], step_fn=T._step_fn)
    graph = [float(step(bt)) for bt in batches]
    # a second eager model: is eager itself reproducible?
    m3, _ = T._fused(dev)
    m3.branch_streams = streams
    m3.optimizer.prepare().set_capturable(True)
    c.zero_()
    eager2 = []
    for bt in batches:
        H.step_advance()
        eager2.append(float(T._step_fn(m3, bt)))
    H.disable_device_step()
    print(tag, "eager", eager, "eager2", eager2, "graph", graph, "max rel", float(np.max(np.abs(np.array(graph) / np.array(eager) - 1))), flush=True)


from applecider_amd import _lib
orig = H.fftconv_covered
scenario("default")
H._FFT_MATH = _lib.MATH_F32
scenario("fft products in f32")
H._FFT_MATH = None
for Lsel in (1024, 256, 64, 16):
    H.fftconv_covered = lambda B, L, Cin, Cout, k, Lsel=Lsel: L == Lsel and orig(B, L, Cin, Cout, k)
    scenario("fft only at L=%d" % Lsel)
