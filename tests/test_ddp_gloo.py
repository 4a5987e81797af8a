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


class _FakeStreamOps:
    """Stands in for ddp._CudaStreamOps on CPU: `current` names the stream the 'kernel wrapper' runs
    on; events remember the stream they were recorded on; the exchange stream logs what it waits for."""

    def __init__(self):
        self.current = "main"
        self.waited = []          # stream names the exchange stream was told to wait for, in order
        self.on_exchange_calls = 0

    def current_key(self):
        return self.current

    def record(self, event=None):
        event = event if event is not None else {}
        event["stream"] = self.current
        return event

    def exchange_wait(self, event):
        self.waited.append(event["stream"])

    def on_exchange(self):
        import contextlib
        self.on_exchange_calls += 1
        return contextlib.nullcontext()

    def join_exchange(self):
        pass


def test_bucket_collective_waits_for_every_writer_stream():
    """A bucket whose parameters got their gradients from kernels on different streams must order
    its collective after ALL of them (the main stream was never waited for)."""
    from applecider_amd.ddp import GradBuckets
    from applecider_amd.optim import FlatParameters
    ps = [torch.nn.Parameter(torch.randn(64)) for _ in range(6)]
    fp = FlatParameters([{"params": ps}])
    fp.flatten()
    ops = _FakeStreamOps()

    class GB(GradBuckets):
        def _all_reduce(self, view):
            self.reduced = getattr(self, "reduced", []) + [(view.data_ptr(), view.numel(), list(ops.waited))]
            return None

    gb = GB(fp, bucket_bytes=4 * 128, stream_ops=ops)      # 2 parameters per bucket -> 3 buckets
    assert len(gb.buckets) == 3
    # bucket 0: spectra branch (main) then photometry stream; bucket 1: image stream only;
    # bucket 2: one parameter from the image stream, one never reports (no gradient)
    for idx, stream in ((0, "main"), (1, "s_pho"), (2, "s_img"), (3, "s_img"), (4, "s_img")):
        ops.current = stream
        gb._report(idx)
    assert [b for b, _ in gb.wait_log] == [0, 1]
    assert gb.wait_log[0] == (0, ("main", "s_pho"))
    assert gb.wait_log[1] == (1, ("s_img",))
    # the collective of bucket 0 was issued from s_pho's hook but waited for main's event as well
    assert set(gb.reduced[0][2]) == {"main", "s_pho"}
    ops.current = "main"
    ops.waited.clear()
    gb.finish()                                            # launches bucket 2 from the main stream
    assert gb.last_wait_log[-1] == (2, ("s_img",))
    assert "s_img" in gb.reduced[2][2] and "main" in gb.reduced[2][2]
    assert ops.on_exchange_calls == 3
    # a second report of the same parameter in one backward (sink + autograd hook) is ignored
    gb._report(0)
    gb._report(0)
    assert gb.duplicates == [0]
    gb.remove()


def test_rank_offsets_dropout_seed():
    from applecider_amd import hipops as H
    torch.manual_seed(1234)
    H.set_seed_offset(0)
    a = H.next_seed()
    H.set_seed_offset(3)
    b = H.next_seed()
    H.set_seed_offset(0)
    assert a != b and (b - a) % (1 << 64) != 0xD1B54A32D192ED03   # not just the counter step


@pytest.mark.parametrize("gpus", [4, 8])
def test_bench_self_launch_dry_run(gpus):
    """`python bench.py --gpus N` without WORLD_SIZE becomes the torch.distributed.run launcher: one rank per GPU,
    rendezvous on 127.0.0.1, and dmabuf IPC (HSA_ENABLE_IPC_MODE_LEGACY=0) in the ranks' environment even when the
    caller's shell did not export it (without it RCCL fails with hipIpcGetMemHandle: invalid argument on this pool)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "7",
                          "--warmup", "2", "--dry-run-launch"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = plan["launch"]
    assert plan["ranks"] == gpus and plan["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert f"--nproc-per-node={gpus}" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", str(gpus), "--steps", "7", "--warmup", "2"]


def test_flat_optimizer_state_dict_round_trip_and_rehoming():
    """optimizer state can be checkpointed (torch.optim layout) and survives re-homing of
    the parameters after training started.  Pure tensor logic: no kernel is launched."""
    from applecider_amd.optim import FlatAdam, FlatSGD
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 64, 7)]
    opt = FlatAdam([{"params": ps[:2], "lr": 1e-3}, {"params": ps[2:], "lr": 5e-4}])
    opt.prepare()
    opt.step_count = 3
    opt.exp_avg.copy_(torch.arange(opt.exp_avg.numel(), dtype=torch.float32))
    opt.exp_avg_sq.copy_(torch.arange(opt.exp_avg_sq.numel(), dtype=torch.float32) * 2)
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][1]["params"] == [2]
    assert sd["state"][1]["exp_avg"].shape == (64,) and float(sd["state"][0]["step"]) == 3.0
    # torch.optim.Adam accepts the same structure for the same parameters
    ref = torch.optim.Adam([{"params": ps[:2]}, {"params": ps[2:]}])
    ref.load_state_dict(sd)
    assert torch.equal(ref.state[ps[1]]["exp_avg"], sd["state"][1]["exp_avg"])
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = FlatAdam([{"params": ps2[:2]}, {"params": ps2[2:]}])
    opt2.load_state_dict(sd)
    assert opt2.step_count == 3 and opt2.param_groups[1]["lr"] == 5e-4
    assert all(torch.equal(a["exp_avg_sq"], b["exp_avg_sq"]) for a, b in
               zip(opt2.state_dict()["state"].values(), sd["state"].values()))
    # re-homing: p.data reassigned after training started -> moments carried over, step count kept
    before = [m.clone() for m, _ in opt._moments_per_param()]
    with torch.no_grad():
        for p in ps:
            p.data = p.data.clone()
    assert not opt.fp.is_current()
    opt._ensure()
    assert opt.fp.is_current() and opt.step_count == 3
    assert all(torch.equal(a, b) for a, (b, _) in zip(before, opt._moments_per_param()))
    sgd = FlatSGD(ps, lr=0.01, momentum=0.9)
    sgd.prepare()
    assert sgd.state_dict()["state"] == {}
    sgd.first = False
    sgd.buf.fill_(1.5)
    ssd = sgd.state_dict()
    sgd2 = FlatSGD([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=0.1)
    sgd2.load_state_dict(ssd)
    assert not sgd2.first and sgd2.param_groups[0]["lr"] == 0.01
    assert float(sgd2.buf[sgd2.fp.offsets[1]]) == 1.5


def test_split_k_rule_by_math_mode():
    """Host logic only: the split-K factor of the weight-gradient products (hipops._split_for) gives the
    split-bf16 kernel (32-deep K tiles) about twice the workgroups of the other modes on long reductions,
    never less than one, and never more workgroups than K tiles allow."""
    from applecider_amd import hipops as H
    shapes = [(512, 1028, 262144), (64, 192, 2097152), (1024, 1536, 8192), (384, 1536, 4608), (128, 384, 524288),
              (512, 128, 66048), (32, 16, 512), (128, 128, 64)]
    try:
        H.set_math("bf16x3")
        x3 = [H._split_for(*s) for s in shapes]
        H.set_math("f32")
        f32 = [H._split_for(*s) for s in shapes]
    finally:
        H.set_math("f32")
    for (m, n, k), a, b in zip(shapes, x3, f32):
        assert a >= 1 and b >= 1
        assert a <= max(1, -(-k // 64) // 4 + 1) * 8          # at least a few K rows per workgroup
        assert a >= b                                         # the split-bf16 kernel never splits less
    assert 1.5 * f32[0] <= x3[0] <= 2.5 * f32[0] and 1.5 * f32[2] <= x3[2] <= 2.5 * f32[2]
    # factors from 6 up are multiples of 8: the K pieces are dealt to the 8 XCDs (ac_gemm.hip map_workgroup)
    assert all(v < 6 or v % 8 == 0 for v in x3 + f32)


def test_public_header_is_plain_c():
    """include/applecider_hip.h is the FFI contract: it must compile as C99 and as C++ on its own (no HIP or
    torch types in any signature)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "applecider_hip.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], check=True)
    subprocess.run(["g++", "-fsyntax-only", "-x", "c++", "-Wall", "-Werror", hdr], check=True)


def test_rccl_log_parser(tmp_path):
    """bench.py, N > 1: rank 0 records RCCL's algorithm / protocol / transport choices (NCCL_DEBUG=INFO into a file) and
    reports them in its JSON line - the parser on the line shapes RCCL prints."""
    import bench
    f = tmp_path / "rank0.log"
    f.write_text("h:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\n"
                 "h:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 7[7] via P2P/IPC\n"
                 "h:1:1 [0] NCCL INFO Connected all rings\nh:1:1 [0] NCCL INFO Connected all trees\n"
                 "h:1:1 [0] NCCL INFO AllReduce: 33554432 Bytes -> Algo 1 proto 2 time 123.4\n"
                 "h:1:1 [0] NCCL INFO AllReduce: 33554432 Bytes -> Algo 1 proto 2 time 123.4\n"
                 "h:1:1 [0] NCCL INFO 32 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 2 p2p channels per peer\n")
    r = bench.parse_rccl_log(str(f))
    assert r["collective_choices"] == {"AllReduce algo 1 proto 2": 2} and r["transports"] == {"P2P/IPC": 2}
    assert r["rings_connected"] and r["trees_connected"] and r["coll_channels"] == 32 and r["log_lines"] == 7
    assert "error" in bench.parse_rccl_log(str(tmp_path / "missing.log"))
