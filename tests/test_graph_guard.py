"""CPU test of the stale-autograd-graph detector behind GraphedTrainStep's capture guard
(applecider_amd/graphstep.py): pure autograd bookkeeping, no GPU needed."""
import torch

from applecider_amd.graphstep import stale_autograd_parameters


class _Sink(torch.autograd.Function):
    """A Function in the style of the package's kernels: the weight is an input, its gradient is written
    elsewhere (returns None)."""

    @staticmethod
    def forward(ctx, x, w):
        return x * w

    @staticmethod
    def backward(ctx, g):
        return g, None


def test_detector_sees_exactly_the_parameters_a_kept_graph_references():
    ps = [torch.nn.Parameter(torch.randn(4)) for _ in range(3)]
    assert stale_autograd_parameters(ps) == []
    loss = (ps[0] * 2).sum() + _Sink.apply(torch.randn(4, requires_grad=True), ps[2]).sum()
    assert stale_autograd_parameters(ps) == [0, 2]
    loss.backward()                       # backward frees buffers, the nodes stay referenced by `loss`
    assert stale_autograd_parameters(ps) == [0, 2]
    kept = loss.detach()
    del loss
    assert stale_autograd_parameters(ps) == []
    assert kept.grad_fn is None


def test_detector_leaves_no_trace_and_ignores_frozen_parameters():
    p, q = torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.randn(3), requires_grad=False)
    y = (p + q).sum()
    assert stale_autograd_parameters([p, q, None, 3]) == [0]
    node = p.view_as(p).grad_fn.next_functions[0][0]
    assert "_ac_capture_probe" not in node.metadata
    assert stale_autograd_parameters([p] * 10, limit=2) == [0, 1]
    del y
