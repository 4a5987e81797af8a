"""hipGraph-captured inference: the replayed graph must reproduce the eager forward
bit for bit, follow weight updates, handle short batches, and agree with the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = {"mode": "all", "p_d_model": 128, "p_n_heads": 8, "p_n_layers": 4, "p_dropout": 0.4,
       "max_len": 257, "num_classes": 5, "hidden_dim": 64, "fusion": "avg", "lr": 1e-3,
       "beta1": 0.9, "beta2": 0.999, "weight_decay": 0.01}


def _batch(n, seed, dev=None):
    from applecider_amd.synthetic import make_batch
    b = make_batch(n, seed=seed)
    return {k: torch.from_numpy(b[k]) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label")}


@pytest.mark.parametrize("math_mode", ["f32", "bf16"])
def test_graph_replay_matches_eager(dev, math_mode):
    from applecider_amd import hipops as H
    from applecider_amd.inference import GraphedClassifier
    from applecider_amd.models.applecider import AppleCider
    H.set_math(math_mode)
    try:
        torch.manual_seed(3)
        net = AppleCider(dict(CFG)).to(dev)
        net.optimizer.prepare()
        gc = GraphedClassifier(net, batch_size=6, use_probabilities=True)
        for seed, n in ((11, 6), (12, 6), (13, 4), (14, 1)):
            b = _batch(n, seed)
            got = gc.predict(b).clone()
            ref = gc.eager(b)
            assert got.shape == (n, 5)
            if n == 6:
                assert torch.equal(got, ref), (got - ref).abs().max()
            else:   # padded replay: other rows differ, these rows must not
                assert torch.allclose(got, ref, rtol=0, atol=1e-6)
            assert torch.allclose(got.sum(1), torch.ones(n, device=dev), atol=1e-5)
        # records: one per alert, numpy payload
        recs = gc.records(_batch(3, 15), ids=["ZTF1", "ZTF2", "ZTF3"])
        assert [r["id"] for r in recs] == ["ZTF1", "ZTF2", "ZTF3"] and recs[0]["tensor"].shape == (5,)
        # a training step between replays is picked up (weights are read through pointers)
        b = _batch(6, 21)
        before = gc.predict(b).clone()
        net.train()
        tb = _batch(6, 22)
        net.train_step(tuple(tb[k].to(dev) for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label")))
        net.eval()
        gc.refresh()
        after = gc.predict(b).clone()
        assert not torch.equal(before, after)
        assert torch.equal(after, gc.eager(b))
        with pytest.raises(ValueError):
            gc.predict(_batch(7, 1))
    finally:
        H.set_math("f32")


def test_graph_logits_match_oracle(dev):
    """fp32 mode, logits (no softmax): the graphed forward against the CPU oracle at 1e-3."""
    from applecider_amd.inference import GraphedClassifier
    from applecider_amd.models.applecider import AppleCider
    from oracle.functional import applecider_forward
    from oracle.weights import closed_form_state_dict
    from applecider_amd.config import default_config
    net = AppleCider(dict(CFG))
    sd = closed_form_state_dict({k: v.shape for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    net = net.to(dev)
    gc = GraphedClassifier(net, batch_size=4, use_probabilities=False)
    b = _batch(4, 5)
    got = gc.predict(b).cpu()
    ocfg = {"p_n_heads": 8, "p_n_layers": 4, "fusion": "avg",
            "kernel_sizes_per_stage": default_config()["model"]["SpectraNet"]["kernel_sizes_per_stage"]}
    with torch.no_grad():
        ref = applecider_forward(sd, *[b[k] for k in ("photometry", "pad_mask", "metadata", "image", "spectra")], ocfg)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 1e-3, err
    assert torch.equal(got.argmax(1), ref.argmax(1))


def test_fp16_inference_b2048_graph(dev):
    """BASELINE configs[4] as stated: inference-only fused forward, batch 2048, fp16 operands
    (libapplecider_hip_f16.so: v_mfma_f32_32x32x16_f16), hipGraph-captured.  Graph == eager bit for bit,
    probabilities sum to 1, a short batch is padded, scores agree with the fp32 matrix-core mode within
    fp16 operand rounding, and backward through the mode is refused."""
    from applecider_amd import hipops as H
    from applecider_amd.inference import GraphedClassifier
    from applecider_amd.models.applecider import AppleCider
    B = 2048
    torch.manual_seed(5)
    net = AppleCider(dict(CFG)).to(dev)
    net.optimizer.prepare()
    b = _batch(B, 31)
    H.set_math("f16")
    try:
        assert H.get_math() == "f16" and H._H16 == torch.float16
        gc = GraphedClassifier(net, batch_size=B, use_probabilities=True)
        got = gc.predict(b).clone()
        ref16 = gc.eager(b)
        assert got.shape == (B, 5)
        assert torch.equal(got, ref16), (got - ref16).abs().max()
        assert torch.allclose(got.sum(1), torch.ones(B, device=dev), atol=1e-5)
        short = {k: v[:100] for k, v in b.items()}
        assert torch.allclose(gc.predict(short), got[:100], rtol=0, atol=1e-6)
        # training through fp16 operands is refused
        x = torch.randn(64, 128, device=dev, requires_grad=True)
        w = torch.nn.Parameter(torch.randn(128, 128, device=dev))
        with pytest.raises(RuntimeError, match="inference only"):
            H.linear(x, w).sum().backward()
    finally:
        H.set_math("f32")
    # against the exact fp32 matrix-core mode on the first 256 alerts
    gc32 = GraphedClassifier(net, batch_size=256, use_probabilities=True)
    ref = gc32.predict({k: v[:256] for k, v in b.items()}).clone()
    diff = (got[:256] - ref).abs().max().item()
    agree = int((got[:256].argmax(1) == ref.argmax(1)).sum())
    print(f"[fp16 inference] max |dp| vs fp32 = {diff:.2e}, argmax agreement {agree}/256")
    assert diff <= 2e-2 and agree >= 250
    H.set_math("bf16")
    try:   # the same library switch back: bf16 mirrors are rebuilt, not reused from the fp16 epoch
        gcb = GraphedClassifier(net, batch_size=256, use_probabilities=True)
        refb = gcb.predict({k: v[:256] for k, v in b.items()}).clone()
        assert (refb - ref).abs().max().item() <= 6e-2
    finally:
        H.set_math("f32")
