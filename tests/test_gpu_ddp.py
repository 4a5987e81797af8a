"""Data-parallel path on the real kernels: two ranks share the one GPU of the test box through the
gloo backend (RCCL refuses two ranks on one device), run the HIP backward with gradient sinks and
bucketed all-reduce from the hooks, and must end with the average of the two per-rank gradients —
checked against a single process that runs both half-batches."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


FUSED_CFG = {"mode": "all", "p_d_model": 128, "p_n_heads": 8, "p_n_layers": 4, "p_dropout": 0.0,
             "max_len": 257, "num_classes": 5, "hidden_dim": 64, "fusion": "avg", "lr": 1e-3}


class _Fused:
    """The 4-modality model behind the interface the worker uses (its encoders run on three HIP
    streams, so a gradient bucket mixes parameters whose backward kernels ran on different streams)."""

    def __init__(self, dev):
        from common import closed_form_sd
        from applecider_amd.models.applecider import AppleCider
        m = AppleCider(dict(FUSED_CFG))
        m.load_state_dict(closed_form_sd(m))
        self.m = m.to(dev).eval()
        self.this_optimizer = self.m.optimizer

    def named_parameters(self):
        return self.m.named_parameters()

    def loss(self, batch):
        from applecider_amd import hipops as H
        return H.cross_entropy_index(self.m(*batch[:5]), batch[5])


class _Image:
    def __init__(self, dev):
        from common import cfg_default, closed_form_sd
        from applecider_amd.models.astrominn import AstroMiNN
        m = AstroMiNN(cfg_default())
        m.load_state_dict(closed_form_sd(m))
        self.m = m.to(dev).eval()
        self.this_optimizer = self.m.this_optimizer

    def named_parameters(self):
        return self.m.named_parameters()

    def loss(self, batch):
        return self.m.this_criterion(self.m(batch), batch[2])


def _make(dev, fused=False):
    return _Fused(dev) if fused else _Image(dev)


def _batch(seed, dev, fused=False):
    from applecider_amd.synthetic import make_batch
    if fused:
        b = make_batch(3, seed=seed)
        return tuple(torch.from_numpy(b[k]).to(dev) for k in
                     ("photometry", "pad_mask", "metadata", "image", "spectra", "label"))
    b = make_batch(8, seed=seed)
    return tuple(torch.from_numpy(b[k]).to(dev) for k in ("metadata", "image", "target"))


def _worker(rank, world, port, q, overlap=True, sinks=True, fused=False):
    try:
        _worker_body(rank, world, port, q, overlap, sinks, fused)
    except BaseException as e:  # the parent would otherwise sit in q.get() until its timeout
        import traceback
        q.put((rank, None, traceback.format_exc()))
        raise


def _worker_body(rank, world, port, q, overlap, sinks, fused):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": "0"})
    import torch.distributed as dist
    from applecider_amd import ddp
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ddp.init_from_env("gloo")
    from applecider_amd import hipops as H
    H.enable_grad_sinks(sinks)
    m = _make(dev, fused)
    opt = m.this_optimizer.prepare()
    ddp.broadcast_parameters(opt.fp)
    gb = ddp.GradBuckets(opt.fp, bucket_bytes=8 << 20, overlap=overlap)
    assert len(gb.buckets) > 3
    batch = _batch(100 + rank, dev, fused)
    opt.zero_grad()
    loss = m.loss(batch)
    loss.backward()
    # every parameter must have reported exactly one completed gradient before finish() (the fused
    # model owns parameters that are not on its path — e.g. the unused photometry class head — whose
    # buckets are launched by finish())
    assert len(gb._seen) == len(opt.fp.params) or fused
    main_key = torch.cuda.current_stream(dev).cuda_stream
    gb.finish()
    # stream ordering of the exchange: every bucket waited for each stream that wrote into
    # it.  The fused model's encoders run on three streams: the spectra branch on the caller's stream,
    # image and photometry on side streams, and 8 MB buckets mix them.
    log = gb.last_wait_log
    assert sorted(b for b, _ in log) == list(range(len(gb.buckets)))
    if fused:
        streams = set(k for _, w in log for k in w)
        assert len(streams) == 3 and main_key in streams, (streams, main_key)
        assert any(len(w) > 1 for _, w in log), "no bucket mixed gradients of two branches"
    else:
        assert all(w in ((main_key,), ()) for _, w in log)
    torch.cuda.synchronize()
    q.put((rank, opt.fp.grad.cpu().numpy().copy(), float(loss.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,sinks,fused", [(True, True, False), (False, True, False), (True, False, False),
                                                 (True, True, True)])
def test_two_ranks_one_gpu_gradient_average(dev, overlap, sinks, fused):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, overlap, sinks, fused)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, g, loss = q.get(timeout=120)
        assert g is not None, f"rank {rank} failed:\n{loss}"
        got[rank] = (g, loss)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: gradient of each half-batch, averaged
    m = _make(dev, fused)
    opt = m.this_optimizer.prepare()
    ref = []
    for r in range(world):
        batch = _batch(100 + r, dev, fused)
        opt.zero_grad()
        m.loss(batch).backward()
        torch.cuda.synchronize()
        ref.append(opt.fp.grad.clone())
    want = ((ref[0] + ref[1]) / 2).cpu().numpy()
    scale = np.abs(want).max()
    for r in range(world):
        err = np.abs(got[r][0] - want)
        i = int(err.argmax())
        names = [n for n, _ in m.named_parameters()]
        import bisect
        pi = bisect.bisect_right(opt.fp.offsets, i) - 1
        # the flat buffer is ordered by optimizer group, not by named_parameters(): report the offset
        tol = 2e-4 if fused else 1e-5   # fused: split-K atomics of the big conv products reorder sums
        assert err.max() <= tol * scale, (f"rank {r}: max err {err.max():.3e} (scale {scale:.3e}) at flat "
                                           f"index {i}, param slot {pi} of {len(opt.fp.offsets)}, "
                                           f"n_bad {(err > tol * scale).sum()}, local-vs-want "
                                           f"{np.abs(ref[r].cpu().numpy() - want).max():.3e}")
    assert np.abs(got[0][0] - got[1][0]).max() == 0.0  # replicas hold identical averaged gradients


def _rccl_worker(port, q):
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0",
                           "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        from applecider_amd import ddp
        from applecider_amd import hipops as H
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1)
        assert ddp.rccl_ranks() == 1
        m = _Fused(dev)
        assert m.m.branch_streams
        opt = m.this_optimizer.prepare()
        ddp.broadcast_parameters(opt.fp)          # world 1: no collective, must not raise
        batch = _batch(100, dev, True)
        # reference gradient: the same step without buckets
        opt.zero_grad()
        m.loss(batch).backward()
        torch.cuda.synchronize()
        want = opt.fp.grad.clone()
        # the bucket path as the N > 1 job runs it: hooks + gradient sinks, exchange stream, async RCCL
        # all-reduce per bucket, h.wait() + stream join + device-side averaging in finish()
        gb = ddp.GradBuckets(opt.fp, bucket_bytes=8 << 20, overlap=True, stream_ops=ddp._CudaStreamOps(dev))
        assert gb._hooks and len(gb.buckets) > 3
        main_key = torch.cuda.current_stream(dev).cuda_stream
        p0 = opt.fp.flat.clone()
        for it in range(2):                       # second pass: events and buckets are re-used
            opt.zero_grad()
            loss = m.loss(batch)
            loss.backward()
            launched_early = sum(gb.launched)
            gb.finish()
            log = gb.last_wait_log
            assert sorted(b for b, _ in log) == list(range(len(gb.buckets)))
            streams = set(k for _, w in log for k in w)
            assert len(streams) == 3 and main_key in streams, (streams, main_key)
            assert any(len(w) > 1 for _, w in log)
            assert launched_early >= 1, "no bucket was launched from the hooks while backward ran"
            # SURVEY 8e / VERDICT r3 next #9: the overlap counts on the LATE layers' buckets leaving first - the
            # 12 M weights of SpectraNet's last stage and ConvNeXt's stage 3 are ready long before the stems
            names = {id(p_): n for n, p_ in m.m.named_parameters()}
            order = {b: i for i, (b, _) in enumerate(log)}

            def buckets(sub, only=False):
                hit = {}
                for i, p_ in enumerate(opt.fp.params):
                    hit.setdefault(gb.bucket_of[i], []).append(sub in names[id(p_)])
                return [b for b, v in hit.items() if (all(v) if only else any(v))]
            for late, early in (("spectra_encoder.all_stages.4.", "spectra_encoder.all_stages.0."),
                                ("image_tower.backbone.stages.3.", "image_tower.backbone.stem.")):
                late_b, early_b = buckets(late, only=True), buckets(early)
                assert late_b and early_b, (late, early)
                assert max(order[b] for b in late_b) < min(order[b] for b in early_b), (late, early, log)
            assert launched_early >= len(buckets("spectra_encoder.all_stages.4.", only=True))
            torch.cuda.synchronize()
            got = opt.fp.grad.clone()
            scale = want.abs().max().item()
            err = (got - want).abs().max().item()
            assert err <= 2e-4 * scale, (it, err, scale)   # split-K atomics reorder sums between two backwards
        opt.step()
        torch.cuda.synchronize()
        assert not torch.equal(opt.fp.flat, p0)
        gb.remove()
        q.put(("ok", ddp.rccl_ranks(), len(gb.buckets), float(loss.item())))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        q.put(("fail", traceback.format_exc(), 0, 0.0))
        raise


def test_one_rank_rccl_through_the_bucket_path(dev):
    """The first RCCL communicator this code meets must not be the 8-GPU run: one rank, backend "nccl"
    (= RCCL), GradBuckets forced onto the exchange-stream path with the three encoder streams on — async
    all-reduce handles under ProcessGroupNCCL semantics, stream-side waits, averaging by 1/1.  Runs in a
    child process (a process group is process-wide state)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    status, a, nb, loss = q.get(timeout=300)
    p.join(timeout=120)
    assert status == "ok", a
    assert a == 1 and nb > 3 and np.isfinite(loss)
    assert p.exitcode == 0
