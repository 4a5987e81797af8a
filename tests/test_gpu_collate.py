"""(f-1) collate -> pinned host -> async H2D on the MI355X (replaces the blocking
`.to(device)` calls of brew_cider.py:991-994 / Time2Vec.py:38-45)."""

import numpy as np
import pytest
import torch

from common import T, gold

pytestmark = pytest.mark.gpu


def _host_batch(seed, B=6):
    from applecider_amd.synthetic import make_batch
    b = make_batch(B, seed=seed)
    return tuple(b[k] for k in ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))


def test_pinned_stager_values_and_slot_reuse(dev):
    from applecider_amd.datasets.collate import PinnedStager
    st = PinnedStager(dev, depth=2)
    hosts = [_host_batch(s) for s in range(5)]
    kept = []
    for h in hosts:                       # 5 batches through 2 slots: every slot is reused twice
        d = st.stage(h)
        assert all(t.is_cuda for t in d)
        kept.append(d)
    torch.cuda.synchronize()
    for h, d in zip(hosts, kept):         # earlier device batches were not overwritten by slot reuse
        for a, t in zip(h, d):
            assert t.dtype == torch.from_numpy(np.asarray(a)).dtype
            assert np.array_equal(t.cpu().numpy(), a)
    assert len(st.host[0]) == 6 and all(buf.is_pinned() for buf in st.host[0].values())
    assert st.bytes_staged == 5 * sum(np.asarray(a).nbytes for a in hosts[0])
    # already-pinned sources are copied from directly (no second host copy)
    pinned = tuple(torch.from_numpy(np.ascontiguousarray(a)).pin_memory() for a in hosts[0])
    d = st.stage(pinned)
    torch.cuda.synchronize()
    assert all(np.array_equal(t.cpu().numpy(), a) for t, a in zip(d, hosts[0]))


def test_pinned_stager_orders_the_consumer_stream(dev):
    """The consumer never reads a batch before its copy landed, on whichever stream it acquires:
    a kernel queued on a side stream right after acquire() sees the new values although the copy stream
    is kept busy by a large copy in front of it (no host synchronisation in between)."""
    from applecider_amd.datasets.collate import PinnedStager
    st = PinnedStager(dev, depth=2)
    big = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()      # ~5 ms of PCIe ahead of the batch
    side = torch.cuda.Stream(device=dev)
    sums = []
    for i in range(4):
        x = np.full((1 << 20,), float(i + 1), dtype=np.float32)
        with torch.cuda.stream(st.copy_stream):
            junk = big.to(dev, non_blocking=True)
        ticket = st.prefetch((x,))
        with torch.cuda.stream(side):
            (xd,) = st.acquire(ticket)
            sums.append(xd.sum())        # queued immediately; must wait for the event, not the host
        del junk
    torch.cuda.synchronize()
    assert [float(s) for s in sums] == [float((i + 1) * (1 << 20)) for i in range(4)]


def test_collate_fused_to_model(dev):
    """sample tuples -> collate_fused (explicit mean/std) -> PinnedStager -> the fused model: the
    logits equal those of the same host arrays moved with blocking copies."""
    from applecider_amd.datasets.collate import PinnedStager, collate_fused
    from applecider_amd.models.applecider import AppleCider
    from applecider_amd.synthetic import make_batch
    from common import closed_form_sd
    g = gold("g11_collate_fused.npz")
    b = make_batch(5, seed=11)
    samples = [(g[f"seq{i}"], b["metadata"][i], b["image"][i], b["spectra"][i], int(b["label"][i]))
               for i in range(5)]
    host = collate_fused(samples, g["mean"], g["std"])
    assert np.abs(host[0] - g["out.photometry"]).max() <= 1e-6 * np.abs(g["out.photometry"]).max()
    assert np.array_equal(host[1], g["out.photo_mask"])
    cfg = {"mode": "all", "p_d_model": 128, "p_n_heads": 8, "p_n_layers": 4, "p_dropout": 0.0,
           "max_len": 257, "num_classes": 5, "hidden_dim": 64, "fusion": "avg", "lr": 1e-3}
    m = AppleCider(cfg)
    m.load_state_dict(closed_form_sd(m))
    m = m.to(dev).eval()
    st = PinnedStager(dev)
    with torch.no_grad():
        staged = st.stage(host)
        y1 = m(*staged[:5])
        y2 = m(*[T(a).to(dev) for a in host[:5]])
    assert torch.equal(y1, y2)
    assert torch.equal(staged[5].cpu(), T(host[5]))


@pytest.mark.parametrize("max_len", [None, 257, 64])
def test_device_side_pad_and_normalise_equals_host_collate_and_reference(dev, max_len):
    """(f-1), VERDICT r3 missing #6: pad / truncate / (x - mean) / (std + 1e-8) on the DEVICE (ac_collate_photometry) from
    the ragged curves, bit-identical to the host numpy collate and - at the reference's own width - to golden g11 = the
    reference's legacy collate (Time2Vec.py:18-45), padding rows standardised like the reference does."""
    from applecider_amd.datasets.collate import PinnedStager, collate_fused, collate_fused_device
    from applecider_amd.synthetic import make_batch
    g = gold("g11_collate_fused.npz")
    b = make_batch(5, seed=11)
    samples = [(g[f"seq{i}"], b["metadata"][i], b["image"][i], b["spectra"][i], int(b["label"][i])) for i in range(5)]
    host = collate_fused(samples, g["mean"], g["std"], max_len=max_len)
    st = PinnedStager(dev)
    devt = collate_fused_device(samples, g["mean"], g["std"], st, max_len=max_len)
    torch.cuda.synchronize()
    assert devt[1].dtype == torch.bool
    for a, t in zip(host, devt):
        assert tuple(t.shape) == a.shape
        assert np.array_equal(t.cpu().numpy(), a), "device collate differs from the host collate"
    if max_len is None:
        assert np.array_equal(devt[1].cpu().numpy(), g["out.photo_mask"])
        ref = g["out.photometry"]
        assert np.abs(devt[0].cpu().numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
    # truncation: curves longer than L are cut, none is read past its end
    if max_len == 64:
        lens = [int(x) for x in g["lens"]]
        assert [int((~devt[1][i]).sum()) for i in range(5)] == [min(n, 64) for n in lens]
