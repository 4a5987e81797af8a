"""Data-parallel gradient exchange for the MI355X path.

The reference has no multi-process code at all; the path shards purely by batch (samples are
independent: LayerNorm/softmax are per-sample, there is no BatchNorm on the default path), so the only
exchange per step is a gradient all-reduce.  One process per GPU, `torch.distributed` backend "nccl"
(= RCCL over xGMI on ROCm); "gloo" on CPU for tests.

Design for MI355X: gradients already live in ONE contiguous fp32 buffer (applecider_amd.optim), so a
bucket is just a slice of it — no packing copies, few large collectives (xGMI rings are per-link
bound, so bigger messages amortise the ~10 us launch/latency floor).  Buckets are cut at parameter
boundaries in buffer order; a bucket is launched from autograd's post-accumulate hooks as soon as
every parameter in it has its gradient, i.e. while the rest of backward is still running.
`finish()` launches whatever is left (parameters that received no gradient), waits, and averages on
the device.

Stream ordering.  The three encoders of the fused model run on three HIP streams, and autograd
replays a backward node on the stream of its forward, so ONE bucket is written from several streams
(bucket boundaries are byte counts, not branch boundaries).  Every gradient report therefore records
an event on the stream that wrote it; the collective of a bucket is issued on a dedicated exchange
stream that first waits for the latest event of every writer stream of that bucket.  No compute stream
ever waits for another branch on behalf of the exchange (the branches keep overlapping), and the
collective cannot start before the last split-K atomic of any branch has landed.
"""

from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class _CudaStreamOps:
    """The three stream primitives the bucket launcher needs (tests substitute a recorder)."""

    def __init__(self, device):
        self.device = device
        self.exchange = torch.cuda.Stream(device=device)

    def current_key(self):
        """Identity of the stream the calling autograd node runs on (hashable)."""
        return torch.cuda.current_stream(self.device).cuda_stream

    def record(self, event=None):
        """(Re-)record `event` on the current stream; returns it."""
        if event is None:
            event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(self.device))
        return event

    def exchange_wait(self, event):
        self.exchange.wait_event(event)

    def on_exchange(self):
        return torch.cuda.stream(self.exchange)

    def join_exchange(self):
        """Current stream waits for everything queued on the exchange stream."""
        torch.cuda.current_stream(self.device).wait_stream(self.exchange)


class GradBuckets:
    def __init__(self, flat_params, process_group=None, bucket_bytes: int = 32 << 20,
                 overlap: bool = True, stream_ops=None):
        """flat_params: applecider_amd.optim.FlatParameters (already flattened).
        stream_ops: object with the _CudaStreamOps interface (default: real HIP streams when the
        gradient buffer is on a GPU, none on CPU)."""
        self.fp = flat_params
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.buckets: List[tuple] = []      # (begin, end) element ranges of the flat grad buffer
        self.bucket_of: List[int] = []      # parameter index -> bucket index
        self.pending: List[int] = []
        self.counts: List[int] = []
        self.handles: list = []
        self.launched: List[bool] = []
        self._hooks = []
        self.ops = stream_ops
        if self.ops is None and self.fp.grad is not None and self.fp.grad.is_cuda:
            self.ops = _CudaStreamOps(self.fp.grad.device)
        self._build(bucket_bytes // 4)
        self._inv_world = None
        self.last_wait_log: List[tuple] = []
        if self.world > 1 or stream_ops is not None:
            self._index = {id(p): i for i, p in enumerate(self.fp.params)}
            for i, p in enumerate(self.fp.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
            # gradients written straight into the flat buffer by the HIP backward kernels
            # (hipops gradient sinks) bypass autograd's accumulation: get told about them too
            try:
                from . import hipops as H
                H._grad_callbacks.append(self._on_sink)
            except Exception:  # CPU-only unit tests without the shared library
                pass

    def _build(self, bucket_elems: int):
        begin, n, cur = 0, 0, 0
        sizes = [self.fp._round(p.numel()) for p in self.fp.params]
        count = 0
        for i, (off, sz) in enumerate(zip(self.fp.offsets, sizes)):
            self.bucket_of.append(cur)
            n += sz
            count += 1
            if n >= bucket_elems or i == len(sizes) - 1:
                self.buckets.append((begin, off + sz))
                self.counts.append(count)
                begin, n, count = off + sz, 0, 0
                cur += 1
        # one reusable event per (bucket, writer stream): re-recorded every step
        self._events: List[Dict[object, object]] = [dict() for _ in self.buckets]
        self.reset()

    def reset(self):
        self._seen = set()
        self.duplicates = []
        self.pending = list(self.counts)
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self._writers: List[set] = [set() for _ in self.buckets]
        self.wait_log = []                  # (bucket, writer-stream keys waited for) of this backward

    def _all_reduce(self, view):
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch(self, b: int):
        if self.launched[b]:
            return
        self.launched[b] = True
        lo, hi = self.buckets[b]
        view = self.fp.grad[lo:hi]
        ops = self.ops
        if ops is None:
            self.handles.append(self._all_reduce(view))
            return
        # the exchange stream waits for the last gradient write of EVERY stream that wrote into this
        # bucket (events recorded at report time), and for the launching stream's work so far (covers
        # parameters that never reported: their slice was zeroed on that stream)
        writers = sorted(self._writers[b], key=repr)
        for key in writers:
            ops.exchange_wait(self._events[b][key])
        ops.exchange_wait(ops.record())
        self.wait_log.append((b, tuple(writers)))
        with ops.on_exchange():
            self.handles.append(self._all_reduce(view))

    def _report(self, idx: int):
        # A parameter is counted once per backward: a gradient written through a sink is reported by
        # the kernel wrapper, and autograd still runs the parameter's post-accumulate hook afterwards
        # (with an undefined gradient) - the first report wins.
        if idx in self._seen:
            self.duplicates.append(idx)
            return
        self._seen.add(idx)
        b = self.bucket_of[idx]
        if self.ops is not None:
            key = self.ops.current_key()
            self._events[b][key] = self.ops.record(self._events[b].get(key))
            self._writers[b].add(key)
        self.pending[b] -= 1
        if self.pending[b] == 0 and self.overlap:
            self._launch(b)

    def _make_hook(self, idx: int):
        def hook(_param):
            self._report(idx)
        return hook

    def _on_sink(self, param):
        i = self._index.get(id(param))
        if i is not None:
            self._report(i)

    def finish(self):
        """Call after backward(): completes the exchange and leaves the AVERAGED gradient in the
        flat buffer.  No host synchronisation on GPU (stream waits only)."""
        # single process without hooks (no process group, no explicit stream_ops): nothing to exchange —
        # also on a GPU, where `ops` exists but neither hooks nor a communicator do
        if self.world == 1 and not self._hooks:
            return
        for b in range(len(self.buckets)):
            self._launch(b)
        self.last_wait_log = list(self.wait_log)
        for h in self.handles:
            if h is not None:
                h.wait()
        if self.ops is not None:
            self.ops.join_exchange()
        g = self.fp.grad
        if g.is_cuda:
            from . import hipops as H
            if self._inv_world is None or self._inv_world.device != g.device:
                self._inv_world = torch.full((1,), 1.0 / self.world, device=g.device)
            H._lib.check(H._lib_().ac_scale_by_dev(H._p(g), g.numel(), H._p(self._inv_world),
                                                   H._stream()), "ac_scale_by_dev")
        else:
            g.mul_(1.0 / self.world)  # gloo / CPU test path
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        try:
            from . import hipops as H
            if self._on_sink in H._grad_callbacks:
                H._grad_callbacks.remove(self._on_sink)
        except Exception:
            pass


def init_from_env(backend: Optional[str] = None):
    """One process per GPU, rendezvous from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun).
    Also offsets this rank's dropout seed stream: every rank calls torch.manual_seed with the same
    value (identical initial weights before the broadcast), but the replicas must not draw identical
    dropout masks."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    try:
        from . import hipops as H
        H.set_seed_offset(rank)
    except Exception:  # CPU-only unit tests without the shared library
        pass
    return rank, local, world


def rccl_ranks() -> int:
    """Number of ranks of the live RCCL (backend "nccl") communicator; 0 when the job does not run
    over RCCL (single process, or a gloo rehearsal)."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return dist.get_world_size()
    return 0


def broadcast_parameters(flat_params, src: int = 0, process_group=None):
    """All ranks start from rank `src`'s weights (one collective over the flat buffer)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_params.flat, src=src, group=process_group)
