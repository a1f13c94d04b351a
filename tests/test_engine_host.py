"""Host-side logic of the training engine (CPU, no kernels): flat buffer layout, all-reduce buckets, gradient
destinations, live runs (parameters without a gradient are skipped like torch.optim does)."""
import torch

from mia_hip import ops
from models.unet import UNet
from training.engine import FlatOptimizer


def _model():
    torch.manual_seed(0)
    return UNet(2, 1, 3, [4, 8, 16], deep_supervision=True, ds_layer=2, normalization="batch", dropout_prob=None)


def test_flat_layout_buckets_and_views():
    m = _model()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    opt = FlatOptimizer(m, "adam", weight_decay=5e-4, bucket_bytes=2048)
    params = [p for p in m.parameters() if p.requires_grad]
    assert opt.params == params[::-1]  # reverse forward order: the first gradients produced sit first
    end = 0
    for p, o in zip(opt.params, opt.offsets):
        assert o % 4 == 0 and o >= end  # 16-byte aligned, no overlap
        assert p.data_ptr() == opt.flat_param.data_ptr() + 4 * o
        end = o + p.numel()
    assert opt.flat_param.numel() >= end
    for k, v in m.state_dict().items():  # values survived the move into the flat buffer
        assert torch.equal(v, before[k]), k
    # buckets tile [0, total) without gaps and end on parameter boundaries
    assert opt.buckets[0][0] == 0 and opt.buckets[-1][1] == opt.flat_param.numel() and len(opt.buckets) > 2
    ends = {o + (p.numel() + 3) // 4 * 4 for p, o in zip(opt.params, opt.offsets)}
    for (s0, e0), (s1, _) in zip(opt.buckets, opt.buckets[1:]):
        assert e0 == s1 and e0 in ends
    for o, bi in zip(opt.offsets, opt.param_bucket):
        assert opt.buckets[bi][0] <= o < opt.buckets[bi][1]


def test_gradient_destinations_and_live_runs():
    m = _model()
    opt = FlatOptimizer(m, "sgd")
    opt.zero_grad()
    assert all(p.grad is None for p in opt.params) and float(opt.flat_grad.abs().max()) == 0.0
    p0, o0 = opt.params[3], opt.offsets[3]
    d1 = ops.grad_dest(p0)
    assert d1.data_ptr() == opt.flat_grad.data_ptr() + 4 * o0 and d1.shape == p0.shape
    assert ops.grad_dest(p0) is None  # the slice is claimed until this accumulation completes (second node -> fresh tensor)
    ops.release_grad_dest(p0)
    d2 = ops.grad_dest(p0)
    assert d1 is not d2 and d1.data_ptr() == d2.data_ptr()
    assert ops.grad_dest(torch.nn.Parameter(torch.zeros(3))) is None  # unregistered parameter
    p0.grad = d1
    assert ops.grad_dest(p0) is None  # a gradient is present: autograd has to accumulate, no direct write
    # live runs: only parameters that received a gradient are stepped
    opt.zero_grad()
    assert opt._live_runs() == []
    for p, o in zip(opt.params, opt.offsets):
        if "decoder.ds" not in [k for k, q in m.named_parameters() if q is p][0]:
            p.grad = opt.flat_grad[o:o + p.numel()].view(p.shape)
    runs = opt._live_runs()
    covered = sum(e - s for s, e in runs)
    want = sum((p.numel() + 3) // 4 * 4 for p in opt.params if p.grad is not None)
    # the unused deep-supervision heads sit at the front of the reverse-order buffer: the live part starts behind them
    assert covered == want and runs[0][0] > 0 and runs[-1][1] == opt.flat_param.numel()
    for p in opt.params:
        p.grad = opt.flat_grad[:p.numel()].view(p.shape)
    assert opt._live_runs() == [(0, opt.flat_param.numel())]
    # a second optimizer takes the parameters over: destinations move, the first one's hooks go quiet
    opt2 = FlatOptimizer(m, "adam")
    opt2.zero_grad()
    d = ops.grad_dest(p0)
    o2 = opt2.offsets[[id(q) for q in opt2.params].index(id(p0))]
    assert d.data_ptr() == opt2.flat_grad.data_ptr() + 4 * o2 and p0._mia_flat_owner is opt2._token


class _MockNode(torch.autograd.Function):
    """Stands in for a backward node of ops.py: writes the parameter gradient into ops.grad_dest(w) when it gets one."""

    @staticmethod
    def forward(ctx, x, w, scale):
        ctx.w, ctx.scale, ctx.n = w, scale, x.numel()
        return (x * w.detach().sum() * scale).sum()

    @staticmethod
    def backward(ctx, g):
        dst = ops.grad_dest(ctx.w)
        val = torch.full_like(ctx.w, ctx.scale)
        if dst is None:
            dst = val
        else:
            dst.copy_(val)
        return None, dst, None


def test_two_nodes_sharing_one_parameter_in_one_backward_accumulate():
    """two autograd nodes that use the same registered parameter inside ONE backward pass (model called twice
    before one loss.backward(), weight sharing) must sum their gradients, not alias the flat slice (2 instead of 11)."""
    m = _model()
    opt = FlatOptimizer(m, "sgd")
    p0, o0 = opt.params[3], opt.offsets[3]
    for _ in range(2):  # the claim is released again by the post-accumulate hook / zero_grad
        opt.zero_grad()
        x = torch.ones(4)
        loss = _MockNode.apply(x, p0, 1.0) + _MockNode.apply(x, p0, 10.0)
        loss.backward()
        assert torch.allclose(p0.grad, torch.full_like(p0, 11.0))
        assert p0.grad.data_ptr() == opt.flat_grad.data_ptr() + 4 * o0  # the flat buffer holds the sum
        assert torch.allclose(opt.flat_grad[o0:o0 + p0.numel()], torch.full((p0.numel(),), 11.0))
    # accumulation over two backward passes without zero_grad still adds
    _MockNode.apply(torch.ones(4), p0, 5.0).backward()
    assert torch.allclose(opt.flat_grad[o0:o0 + p0.numel()], torch.full((p0.numel(),), 16.0))
