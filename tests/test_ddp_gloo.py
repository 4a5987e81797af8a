"""Data-parallel exchange on CPU: two gloo ranks, flat-buffer gradient buckets launched from autograd
hooks, result = average of the per-rank gradients (applecider_amd/ddp.py).  The model here is a
plain torch graph: only the exchange logic is under test, not the HIP kernels."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bucket_bytes, overlap, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    from applecider_amd.ddp import GradBuckets, broadcast_parameters, init_from_env
    from applecider_amd.optim import FlatParameters
    r, _, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different init per rank -> broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(16, 33), torch.nn.Tanh(), torch.nn.Linear(33, 7),
                              torch.nn.Linear(7, 3))
    unused = torch.nn.Parameter(torch.ones(5))  # never receives a gradient
    fp = FlatParameters([{"params": list(net.parameters()) + [unused]}])
    fp.flatten()
    broadcast_parameters(fp)
    gb = GradBuckets(fp, bucket_bytes=bucket_bytes, overlap=overlap)
    import copy
    ref_net = copy.deepcopy(net)  # same (broadcast) weights, no hooks: the purely local gradient
    outs = []
    for step in range(2):
        fp.zero_grad()
        ref_net.zero_grad()
        gen = torch.Generator().manual_seed(1000 * rank + step)
        x = torch.randn(8, 16, generator=gen)
        net(x).square().mean().backward()
        ref_net(x).square().mean().backward()
        gb.finish()
        local = torch.cat([p.grad.reshape(-1) for p in ref_net.parameters()])
        avg = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        assert unused.grad.abs().max() == 0
        outs.append((local.numpy().copy(), avg.numpy().copy(), fp.flat.detach().numpy().copy()))
    q.put((rank, len(gb.buckets), outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,overlap", [(256, True), (1 << 20, True), (512, False)])
def test_two_rank_gradient_average(bucket_bytes, overlap):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, nb, outs = q.get(timeout=120)
        res[rank] = (nb, outs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if bucket_bytes == 256:
        assert res[0][0] > 1  # several buckets exercised
    for step in range(2):
        l0, a0, w0 = (torch.from_numpy(t) for t in res[0][1][step])
        l1, a1, w1 = (torch.from_numpy(t) for t in res[1][1][step])
        assert torch.equal(w0, w1)  # broadcast made the replicas identical
        want = (l0 + l1) / 2
        assert torch.allclose(a0, want, atol=1e-7) and torch.allclose(a1, want, atol=1e-7)
        assert not torch.allclose(l0, l1)


def test_bucket_layout_covers_buffer():
    from applecider_amd.ddp import GradBuckets
    from applecider_amd.optim import FlatParameters
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 100, 64, 1, 300)]
    fp = FlatParameters([{"params": ps[:2]}, {"params": ps[2:]}])
    fp.flatten()
    gb = GradBuckets(fp, bucket_bytes=4 * 128)
    assert gb.buckets[0][0] == 0 and gb.buckets[-1][1] == fp.flat.numel()
    for (a, b), (c, d) in zip(gb.buckets, gb.buckets[1:]):
        assert b == c and a < b
    assert sum(gb.counts) == len(ps)
    # parameters stay views of the flat buffer and keep their values
    for p, off in zip(fp.params, fp.offsets):
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * off
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * off
    assert fp.is_current()
