"""Host logic of ops.ParamGradSink (no GPU: the buffers live wherever `opt` says): which backward pass of an optimiser step
writes and which accumulates, the clearing of a branch that joins late, and the hand-over of the views to `p.grad`."""
import torch

from piml_amd import ops


def test_first_pass_writes_later_passes_accumulate_and_grads_are_assigned():
    sink = ops.ParamGradSink()
    opt = dict(device='cpu', dtype=torch.float32)
    w = [torch.nn.Parameter(torch.zeros(4)) for _ in range(2)]
    keys = [('ped',), ('obs',)]
    with sink.step():
        assert ops.ParamGradSink._active is sink
        bufs, acc = sink.take(keys[:1], 4, opt)                 # the last frame reaches the loss through one branch only
        assert not acc and len(bufs) == 1
        bufs[0].fill_(1.0)
        sink.give(w[0], bufs[0])
        bufs2, acc2 = sink.take(keys, 4, opt)                   # an earlier frame: both branches; the newcomer is cleared
        assert acc2 and bufs2[0] is bufs[0] and float(bufs2[1].abs().sum()) == 0.0
        bufs2[0].add_(2.0); bufs2[1].add_(5.0)
        sink.give(w[0], bufs2[0]); sink.give(w[1], bufs2[1])    # (the first hand-over of a parameter stands)
        _, acc3 = sink.take(keys, 4, opt)
        assert acc3
    assert ops.ParamGradSink._active is None
    assert torch.equal(w[0].grad, torch.full((4,), 3.0)) and torch.equal(w[1].grad, torch.full((4,), 5.0))
    with sink.step():                                           # the next step starts over, in the same buffers
        again, acc = sink.take(keys, 4, opt)
        assert not acc and again[0] is bufs[0]
    existing = torch.nn.Parameter(torch.zeros(4))
    existing.grad = torch.ones(4)
    with sink.step():                                           # a gradient autograd produced on another path is kept
        b, _ = sink.take([('x',)], 4, opt)
        b[0].fill_(2.0)
        sink.give(existing, b[0])
    assert torch.equal(existing.grad, torch.full((4,), 3.0))


def test_steps_do_not_nest():
    sink = ops.ParamGradSink()
    with sink.step():
        try:
            with ops.ParamGradSink().step():
                raise AssertionError('nested step() accepted')
        except RuntimeError:
            pass
    assert ops.ParamGradSink._active is None
