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


def _make(dev):
    from common import cfg_default, closed_form_sd
    from applecider_amd.models.astrominn import AstroMiNN
    m = AstroMiNN(cfg_default())
    m.load_state_dict(closed_form_sd(m))
    return m.to(dev).eval()


def _batch(seed, dev):
    from applecider_amd.synthetic import make_batch
    b = make_batch(8, seed=seed)
    return tuple(torch.from_numpy(b[k]).to(dev) for k in ("metadata", "image", "target"))


def _worker(rank, world, port, q, overlap=True, sinks=True):
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
    m = _make(dev)
    opt = m.this_optimizer.prepare()
    ddp.broadcast_parameters(opt.fp)
    gb = ddp.GradBuckets(opt.fp, bucket_bytes=8 << 20, overlap=overlap)
    assert len(gb.buckets) > 3
    batch = _batch(100 + rank, dev)
    opt.zero_grad()
    loss = m.this_criterion(m(batch), batch[2])
    loss.backward()
    # every parameter must have reported exactly one completed gradient before finish()
    assert len(gb._seen) == len(opt.fp.params)
    gb.finish()
    torch.cuda.synchronize()
    q.put((rank, opt.fp.grad.cpu().numpy().copy(), float(loss.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,sinks", [(True, True), (False, True), (True, False)])
def test_two_ranks_one_gpu_gradient_average(dev, overlap, sinks):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, overlap, sinks)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, g, loss = q.get(timeout=300)
        got[rank] = (g, loss)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: gradient of each half-batch, averaged
    m = _make(dev)
    opt = m.this_optimizer.prepare()
    ref = []
    for r in range(world):
        batch = _batch(100 + r, dev)
        opt.zero_grad()
        m.this_criterion(m(batch), batch[2]).backward()
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
        assert err.max() <= 1e-5 * scale, (f"rank {r}: max err {err.max():.3e} (scale {scale:.3e}) at flat "
                                           f"index {i}, param slot {pi} of {len(opt.fp.offsets)}, "
                                           f"n_bad {(err > 1e-5 * scale).sum()}, local-vs-want "
                                           f"{np.abs(ref[r].cpu().numpy() - want).max():.3e}")
    assert np.abs(got[0][0] - got[1][0]).max() == 0.0  # replicas hold identical averaged gradients
