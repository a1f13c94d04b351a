"""Kernel-level parity: every libmia_hip entry point against torch-CPU fp32 math on the same seeded
inputs (run on the GPU box: pytest -m gpu).  Calls go through the C ABI (ctypes)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def _has_experiments():
    try:
        import mia_hip
        return hasattr(mia_hip.lib(), "mia_conv_mma_cr")
    except Exception:
        return False


# round-4 experiments that left the shipping library in round 5 (column-reduce epilogue): their tests run against probe builds only
# (hipcc -DMIA_EXPERIMENTS, MIA_HIP_LIB=...; tools/r5_store_hazard.sh)
_needs_experiments = pytest.mark.skipif(not _has_experiments(), reason="probe build of libmia_hip (-DMIA_EXPERIMENTS) only")


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def relerr(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def nhwc(x, dtype, dev):  # NCHW cpu -> NHWC device
    return x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)


def nchw(x):  # NHWC device -> NCHW cpu float
    return x.float().cpu().permute(0, 3, 1, 2)


def q(x, dtype):
    """Quantise reference inputs to the kernel's storage dtype so only accumulation order differs."""
    return x.to(dtype).float()


CONV_CASES = [
    # n, cin1, cin2, cout, h, w
    (2, 16, 0, 16, 16, 16),
    (1, 64, 0, 64, 32, 32),
    (2, 8, 0, 24, 20, 28),
    (2, 1, 0, 8, 16, 16),
    (1, 32, 32, 32, 24, 16),
    (1, 12, 20, 40, 9, 13),
    (1, 128, 0, 80, 8, 8),
    (3, 5, 0, 7, 7, 5),
    # branch-free fast path with ragged tiles, two sources, several output-channel blocks, odd sizes
    (2, 64, 64, 64, 40, 72),
    (1, 128, 0, 256, 24, 40),
    (1, 256, 0, 128, 17, 33),
    (2, 32, 0, 96, 30, 18),
    # concatenated inputs / split gradients whose boundary is not a multiple of the 64-channel block
    (1, 96, 96, 96, 24, 40),
    (2, 40, 24, 64, 20, 36),
    (1, 32, 96, 128, 16, 48),
    # 64 -> 64 persistent register-weight kernel (conv64.hip): ragged right / bottom tiles, fewer tiles than workgroups,
    # more tiles than workgroups (several steps of the tile walk)
    (2, 64, 0, 64, 37, 53),
    (1, 64, 0, 64, 9, 200),
    (3, 64, 0, 64, 200, 216),
    (2, 32, 32, 64, 40, 56),  # input gradient = 64 -> (32 | 32): stays on the generic kernel
    (2, 64, 64, 64, 37, 53),  # input gradient = 64 -> (64 | 64): one pass of conv64_dma_kernel<128>, ragged tiles
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_fwd_dgrad_wgrad(case, stride, dtype):
    from mia_hip import ops, CONV_G3S1, CONV_G3S2, CONV_T3S2, WGRAD_3S1, WGRAD_3S2
    dev = _dev()
    n, c1, c2, cout, h, w = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + stride)
    cin = c1 + c2
    x = q(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
    dy = q(torch.randn(n, cout, ho, wo, generator=g), dtype)
    wq = q(wt, dtype)
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, stride=stride, padding=1)
    yr.backward(dy)

    x1 = nhwc(x[:, :c1], dtype, dev)
    x2 = nhwc(x[:, c1:], dtype, dev) if c2 else None
    wd = wt.to(dev)
    pc = ops.PackCache()
    wp, npad, kpad = pc.get(wd, ops._dt(dtype), True)
    mode = CONV_G3S2 if stride == 2 else CONV_G3S1
    y, _, stats = ops.conv_mma(mode, x1, x2, wp, npad, kpad, False, b.to(dev), cout, (ho, wo), want_stats=True)
    torch.cuda.synchronize()
    assert relerr(nchw(y), yr) < TOL[dtype]
    # epilogue statistics: per-(n, c) sum and sum of squares
    s = stats.sum(1).cpu()
    assert relerr(s[..., 0], yr.detach().sum((2, 3))) < 1e-3 + TOL[dtype]
    assert relerr(s[..., 1], (yr.detach() ** 2).sum((2, 3))) < 1e-3 + TOL[dtype]
    # input gradient (two destinations when the input was two sources)
    dyd = nhwc(dy, dtype, dev)
    wb, npb, kpb = pc.get(wd, ops._dt(dtype), False)
    split = c1 if c2 else None
    if stride == 2:
        dx1, dx2, _ = ops.conv_mma(CONV_T3S2, dyd, None, wb, npb, kpb, False, None, cin, (h, w), out_split=split)
    else:
        dx1, dx2, _ = ops.conv_mma(CONV_G3S1, dyd, None, wb, npb, kpb, True, None, cin, (h, w), out_split=split)
    dx = nchw(dx1) if dx2 is None else torch.cat([nchw(dx1), nchw(dx2)], 1)
    assert relerr(dx, xr.grad) < TOL[dtype]
    # weight gradient
    dw = ops.conv_wgrad(WGRAD_3S2 if stride == 2 else WGRAD_3S1, x1, x2, dyd, wt.shape, cout, cin)
    assert relerr(dw, wr.grad) < TOL[dtype]
    # bias gradient
    assert relerr(ops.colsum(dyd), dy.sum((0, 2, 3))) < 1e-3 + TOL[dtype]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 96, 0, 96, 24, 40), (1, 96, 96, 96, 19, 33), (1, 192, 0, 96, 12, 52), (3, 96, 0, 192, 9, 17), (1, 96, 96, 96, 4, 16),
                                  (1, 288, 0, 96, 11, 21), (2, 96, 0, 96, 3, 5)])
def test_weight_gradient_on_96_wide_blocks(case):
    """`wgrad_bf16_dma96_kernel` (csrc/conv_wgrad.hip): 3x3 stride-1 bf16 layers whose channel counts are multiples of 96 and not of 64
    (cfg5's level 0) run ONE 96 x 96 block per pixel tile on the LDS-DMA ring (768 threads, 192-byte-row images) where the 64-wide kernel
    needs 2 x 2 blocks.  One and two sources, several blocks per dimension, ragged and tiny images (tiles hanging over every border);
    bf16 inputs quantised on both sides, so the result differs from the fp32 reference by summation order only.  `mia_wgrad_plan` must
    report the blocks / target / tile height the launch then uses."""
    import ctypes
    from mia_hip import ops, BF16, WGRAD_3S1
    dev = _dev()
    n, c1, c2, cout, h, w = case
    cin = c1 + c2
    g = torch.Generator().manual_seed(cin + cout + h)
    dt = torch.bfloat16
    x = q(torch.randn(n, cin, h, w, generator=g), dt)
    dy = q(torch.randn(n, cout, h, w, generator=g), dt)
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(x, wt, None, stride=1, padding=1).backward(dy)
    x1 = nhwc(x[:, :c1], dt, dev)
    x2 = nhwc(x[:, c1:], dt, dev) if c2 else None
    dw = ops.conv_wgrad(WGRAD_3S1, x1, x2, nhwc(dy, dt, dev), wt.shape, cout, cin)
    assert relerr(dw, wt.grad) < 2e-3
    pb, pt, ph = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ops.call("mia_wgrad_plan", WGRAD_3S1, BF16, c1, c2, cout, -(-cout // 64) * 64, h, ctypes.byref(pb), ctypes.byref(pt), ctypes.byref(ph))
    assert (pb.value, pt.value, ph.value) == (-(-cout // 96) * (-(-c1 // 96) + -(-c2 // 96)), 256, 4)


# Every distinct 3x3 conv of the benchmarked models at their own widths (the golden UNets are 4..20
# channels wide and never reach the branch-free kernels): [64..1024] (cfg2 / cfg3) on the 128x128-input pyramid, encoder
# (one source; stride 2 = first block of the next level), decoder (two sources = skip | upsampled), plus cfg5's
# [96..3072] extremes.  Inputs are quantised to the storage dtype on both sides, so bf16 differs from the fp32-CPU
# reference only by fp32 accumulation order and the final rounding of a bf16 OUTPUT (<= 2^-9 relative per element);
# weight gradients are fp32 on both sides.
WIDE_CASES = [
    (2, 64, 0, 64, 128, 128), (2, 64, 0, 128, 128, 128), (2, 128, 0, 128, 64, 64), (1, 128, 0, 256, 64, 64),
    (1, 256, 0, 256, 32, 32), (1, 256, 0, 512, 32, 32), (1, 512, 0, 512, 16, 16), (1, 512, 0, 1024, 16, 16),
    (1, 1024, 0, 1024, 8, 8), (1, 512, 512, 512, 16, 16), (1, 256, 256, 256, 32, 32), (1, 128, 128, 128, 64, 64),
    (2, 64, 64, 64, 128, 128), (1, 96, 0, 96, 48, 80), (1, 96, 96, 96, 48, 48), (1, 1536, 0, 3072, 6, 6),
    (1, 3072, 0, 3072, 3, 3), (1, 1536, 1536, 1536, 6, 6),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", WIDE_CASES)
def test_conv3x3_benchmark_widths(case, dtype):
    from mia_hip import ops, CONV_G3S1, CONV_G3S2, CONV_T3S2, WGRAD_3S1, WGRAD_3S2
    dev = _dev()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    n, c1, c2, cout, h, w = case
    stride = 2 if (c2 == 0 and cout == 2 * c1) else 1  # channel doubling = first block of a level (unet.py:54-66)
    g = torch.Generator().manual_seed(c1 + cout + h)
    cin = c1 + c2
    x = q(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
    dy = q(torch.randn(n, cout, ho, wo, generator=g), dtype)
    xr, wr = x.clone().requires_grad_(True), q(wt, dtype).clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, stride=stride, padding=1)
    yr.backward(dy)
    x1 = nhwc(x[:, :c1], dtype, dev)
    x2 = nhwc(x[:, c1:], dtype, dev) if c2 else None
    wd = wt.to(dev)
    pc = ops.PackCache()
    wp, npad, kpad = pc.get(wd, ops._dt(dtype), True)
    y, _, stats = ops.conv_mma(CONV_G3S2 if stride == 2 else CONV_G3S1, x1, x2, wp, npad, kpad, False, b.to(dev), cout, (ho, wo),
                               want_stats=True)
    otol = 2e-5 if dtype == torch.float32 else 6e-3  # bf16: output rounding 2^-9 of |y| <= max|y|, + accumulation order
    assert relerr(nchw(y), yr) < otol
    s = stats.sum(1).cpu()
    assert relerr(s[..., 0], yr.detach().sum((2, 3))) < 1e-3
    assert relerr(s[..., 1], (yr.detach() ** 2).sum((2, 3))) < 1e-3
    dyd = nhwc(dy, dtype, dev)
    wb, npb, kpb = pc.get(wd, ops._dt(dtype), False)
    split = c1 if c2 else None
    if stride == 2:
        dx1, dx2, _ = ops.conv_mma(CONV_T3S2, dyd, None, wb, npb, kpb, False, None, cin, (h, w), out_split=split)
    else:
        dx1, dx2, dst = ops.conv_mma(CONV_G3S1, dyd, None, wb, npb, kpb, True, None, cin, (h, w), want_stats=True, out_split=split)
        # epilogue statistics of the input gradient (its per-channel sum is a transposed conv's bias gradient, ops.py)
        assert relerr(dst.sum(1).cpu()[..., 0], xr.grad.sum((2, 3))) < 2e-3
    dx = nchw(dx1) if dx2 is None else torch.cat([nchw(dx1), nchw(dx2)], 1)
    assert relerr(dx, xr.grad) < otol
    dw = ops.conv_wgrad(WGRAD_3S2 if stride == 2 else WGRAD_3S1, x1, x2, dyd, wt.shape, cout, cin)
    assert relerr(dw, wr.grad) < 1e-4  # fp32 slabs of exact products: accumulation order only, in both dtypes


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 32, 16, 8, 8), (1, 64, 32, 16, 16), (2, 20, 12, 5, 9), (1, 256, 128, 4, 4),
                                  (1, 1024, 512, 8, 8), (1, 512, 256, 16, 16), (2, 128, 64, 64, 64), (1, 3072, 1536, 3, 3)])
def test_conv_transpose2x2(case, dtype):
    from mia_hip import ops
    dev = _dev()
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(5)
    x = q(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = torch.randn(cin, cout, 2, 2, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    dy = q(torch.randn(n, cout, 2 * h, 2 * w, generator=g), dtype)
    xr, wr = x.clone().requires_grad_(True), q(wt, dtype).clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    yr.backward(dy)
    xd = nhwc(x, dtype, dev).requires_grad_(True)
    wd = wt.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.ConvTranspose2x2Fn.apply(xd, wd, bd)
    y.backward(nhwc(dy, dtype, dev))
    torch.cuda.synchronize()
    assert relerr(nchw(y), yr) < TOL[dtype]
    assert relerr(nchw(xd.grad), xr.grad) < TOL[dtype]
    assert relerr(wd.grad, wr.grad) < TOL[dtype]
    assert relerr(bd.grad, br.grad) < 1e-3 + TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("stride,c2", [(1, 0), (2, 0), (1, 8)])
def test_plain_block_fwd_bwd(norm, stride, c2, dtype):
    """Fused block vs conv2d -> (dropout mask) -> instance/batch norm -> leaky_relu on CPU."""
    from mia_hip import ops, NORM_BATCH, NORM_INSTANCE
    dev = _dev()
    n, c1, cout, h, w = 3, 8, 24, 20, 12
    cin = c1 + c2
    g = torch.Generator().manual_seed(17)
    x = q(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = q(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9), dtype)
    b = torch.randn(cout, generator=g) * 0.1
    gamma = 1 + 0.2 * torch.randn(cout, generator=g)
    beta = 0.2 * torch.randn(cout, generator=g)
    keep = 0.75
    drop = torch.bernoulli(torch.full((n, cout), keep), generator=g) / keep
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
    dz = q(torch.randn(n, cout, ho, wo, generator=g), dtype)
    leaves = [t.clone().requires_grad_(True) for t in (x, wt, b, gamma, beta)]
    xr, wr, br, gr, ber = leaves
    y = F.conv2d(xr, wr, br, stride=stride, padding=1) * drop[:, :, None, None]
    rm, rv = torch.zeros(cout), torch.ones(cout)
    if norm == "instance":
        yn = F.instance_norm(y, None, None, gr, ber, True, 0.1, 1e-5)
    else:
        yn = F.batch_norm(y, rm, rv, gr, ber, True, 0.1, 1e-5)
    zr = F.leaky_relu(yn, 0.01)
    zr.backward(dz)

    x1 = nhwc(x[:, :c1], dtype, dev).requires_grad_(True)
    x2 = nhwc(x[:, c1:], dtype, dev).requires_grad_(True) if c2 else None
    pd = [t.to(dev).requires_grad_(True) for t in (wt, b, gamma, beta)]
    rmd, rvd, nbt = torch.zeros(cout, device=dev), torch.ones(cout, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    cfg = ops.NormCfg(NORM_BATCH if norm == "batch" else NORM_INSTANCE, True, 1e-5, 0.1,
                      rmd if norm == "batch" else None, rvd if norm == "batch" else None, nbt if norm == "batch" else None,
                      drop.to(dev))
    z = ops.PlainBlockFn.apply(x1, x2, pd[0], pd[1], pd[2], pd[3], stride, cfg)
    z.backward(nhwc(dz, dtype, dev))
    torch.cuda.synchronize()
    tol = TOL[dtype] * 2
    assert relerr(nchw(z), zr) < tol
    dx = nchw(x1.grad) if x2 is None else torch.cat([nchw(x1.grad), nchw(x2.grad)], 1)
    assert relerr(dx, xr.grad) < tol * 2
    assert relerr(pd[0].grad, wr.grad) < tol * 2
    assert relerr(pd[2].grad, gr.grad) < tol * 2
    assert relerr(pd[3].grad, ber.grad) < tol * 2
    # conv bias in front of a norm layer: analytically 0 for instance norm (rounding noise in the reference),
    # non-zero for batch norm with per-sample dropout masks -> compare with an absolute tolerance
    bscale = max(1.0, br.grad.abs().max().item(), dz.abs().sum().sqrt().item())
    assert (pd[1].grad.cpu() - br.grad).abs().max().item() < (2e-2 if dtype == torch.bfloat16 else 2e-5) * bscale
    if norm == "batch":
        assert relerr(rmd, rm) < 1e-4 + TOL[dtype] and relerr(rvd, rv) < 1e-4 + TOL[dtype] and int(nbt.item()) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k1,c0", [(3, 16), (4, 64), (2, 12), (8, 32), (3, 64), (3, 128), (2, 32), (4, 32)])
def test_head(k1, c0, dtype):
    from mia_hip import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    n, h, w = 2, 12, 20
    x = q(torch.randn(n, c0, h, w, generator=g), dtype)
    wt = torch.randn(k1, c0, 1, 1, generator=g) / math.sqrt(c0)
    b = torch.randn(k1, generator=g)
    dl = torch.randn(n, k1, h, w, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, wt, b)]
    yr = F.conv2d(*leaves)
    yr.backward(dl)
    xd = nhwc(x, dtype, dev).requires_grad_(True)
    wd, bd = wt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ops.HeadFn.apply(xd, wd, bd)
    assert y.shape == (n, k1, h, w) and y.dtype == torch.float32
    y.backward(dl.to(dev))  # NCHW-contiguous upstream gradient (stride path)
    torch.cuda.synchronize()
    assert relerr(y, yr) < 1e-5
    assert relerr(nchw(xd.grad), leaves[0].grad) < TOL[dtype]
    assert relerr(wd.grad, leaves[1].grad) < 1e-4
    assert relerr(bd.grad, leaves[2].grad) < 1e-4


def test_losses_golden(golden_dir):
    """HIP Dice / CE / Dice+CE against vectors produced by the reference's own loss classes."""
    import os
    from losses.compound_losses import DiceAndCELoss
    from losses.dice_loss import DiceLoss
    from losses.ce_loss import RobustCrossEntropyLoss
    dev = _dev()
    d = dict(np.load(os.path.join(golden_dir, "losses.npz")))
    logits, labels = torch.from_numpy(d["logits"]).to(dev), torch.from_numpy(d["labels"]).to(dev)
    for layout in ("nchw", "nhwc"):
        for do_bg in (False, True):
            for batch in (False, True):
                for squared in (False, True):
                    key = f"dice_bg{int(do_bg)}_b{int(batch)}_s{int(squared)}"
                    li = logits.clone()
                    if layout == "nhwc":
                        li = li.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
                    li.requires_grad_(True)
                    v = DiceLoss(3, do_bg=do_bg, batch=batch, squared=squared)(li, labels)
                    v.backward()
                    assert abs(v.item() - float(d[key])) < 2e-6, key
                    np.testing.assert_allclose(li.grad.cpu().numpy(), d[key + "_grad"], atol=2e-7, err_msg=key)
    li = logits.clone().requires_grad_(True)
    ce = RobustCrossEntropyLoss()(li, labels[:, None])
    ce.backward()
    assert abs(ce.item() - float(d["ce"])) < 2e-6
    np.testing.assert_allclose(li.grad.cpu().numpy(), d["ce_grad"], atol=2e-7)
    li = logits.clone().requires_grad_(True)
    comp = DiceAndCELoss(dice_kwargs=dict(num_classes=3, do_bg=True))
    v = comp(li, labels, dice_weight=0.7, ce_weight=0.3)
    v.backward()
    assert abs(v.item() - float(d["dice_ce_w"])) < 2e-6
    np.testing.assert_allclose(li.grad.cpu().numpy(), d["dice_ce_w_grad"], atol=2e-7)
    assert abs(comp(logits, labels, dice_weight=0.0, ce_weight=None).item() - float(d["dice_ce_zero_weight_quirk"])) < 2e-6
    # dense (soft) targets of the logits' shape (dice_loss.py:40-41; torch CE with class probabilities)
    soft = torch.from_numpy(d["soft_targets"]).to(dev)
    for squared in (False, True):
        li = logits.clone().requires_grad_(True)
        v = DiceLoss(3, do_bg=False, squared=squared)(li, soft)
        v.backward()
        assert abs(v.item() - float(d[f"dense_dice_s{int(squared)}"])) < 2e-6
        np.testing.assert_allclose(li.grad.cpu().numpy(), d[f"dense_dice_s{int(squared)}_grad"], atol=2e-7)
    li = logits.clone().requires_grad_(True)
    v = DiceAndCELoss(dice_kwargs=dict(num_classes=3, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)(li, soft)
    v.backward()
    assert abs(v.item() - float(d["dense_dice_ce"])) < 2e-6
    np.testing.assert_allclose(li.grad.cpu().numpy(), d["dense_dice_ce_grad"], atol=2e-7)
    # known answers
    lab = torch.tensor([[[0, 1], [2, 2]]], device=dev)
    fn = DiceLoss(2, do_bg=True)
    assert abs(fn(torch.zeros(1, 3, 2, 2, device=dev), lab).item() - 0.6761878354) < 1e-6
    perfect = F.one_hot(lab, 3).permute(0, 3, 1, 2).float() * 100
    assert abs(fn(perfect, lab).item()) < 1e-6
    assert abs(comp.get_ce_loss(torch.zeros(1, 3, 2, 2, device=dev), lab).item() - math.log(3)) < 1e-6
    # retain_graph=True (BADGE selector, badge_selector.py:25)
    li = logits.clone().requires_grad_(True)
    v = comp(li, labels)
    v.backward(retain_graph=True)
    g1 = li.grad.clone()
    v.backward()
    np.testing.assert_allclose(li.grad.cpu().numpy(), 2 * g1.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("k1", [2, 3, 4])
@pytest.mark.parametrize("hw", [(64, 64), (96, 136), (31, 20)])
@pytest.mark.parametrize("flags", [(True, False, False), (False, True, False), (True, False, True)])
def test_dice_ce_vectorised_path_vs_oracle(k1, hw, flags):
    """Channels-last logits (what the UNet head emits) take the 4-pixels-per-thread kernels; every DiceLoss flag
    combination and ragged slab ends against the CPU oracle (oracle/losses_ref.py; tolerance = the golden test's)."""
    from losses.compound_losses import DiceAndCELoss
    from oracle import losses_ref
    dev = _dev()
    do_bg, batch, squared = flags
    g = torch.Generator().manual_seed(100 * k1 + hw[0])
    b = 3
    logits = torch.randn(b, k1, *hw, generator=g) * 2
    labels = torch.randint(0, k1, (b, *hw), generator=g)
    ref_in = logits.clone().requires_grad_(True)
    ref = losses_ref.dice_and_ce(ref_in, labels, k1 - 1, 0.6, 0.9, do_bg=do_bg, batch=batch, squared=squared)
    ref.backward()
    li = logits.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)  # channels-last storage
    fn = DiceAndCELoss(dice_kwargs=dict(num_classes=k1 - 1, do_bg=do_bg, batch=batch, squared=squared), ce_loss=torch.nn.CrossEntropyLoss)
    v = fn(li, labels.to(dev), dice_weight=0.6, ce_weight=0.9)
    v.backward()
    assert abs(v.item() - ref.item()) < 2e-6
    np.testing.assert_allclose(li.grad.cpu().numpy(), ref_in.grad.numpy(), atol=2e-7)


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_out_of_range_label_poisons_loss_and_is_reported(layout):
    """The reference raises for a label outside [0, K1) (scatter index error, dice_loss.py:25-30; CE target bound check).
    The kernels return NaN loss and NaN gradients for such a batch and ops.check_labels() raises."""
    import mia_hip
    from mia_hip import ops
    from losses.compound_losses import DiceAndCELoss
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(2, 3, 16, 16, generator=g).to(dev)
    if layout == "nhwc":
        logits = logits.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    logits.requires_grad_(True)
    labels = torch.randint(0, 3, (2, 16, 16), generator=g).to(dev)
    fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    ok = fn(logits, labels)
    assert math.isfinite(ok.item())
    ops.check_labels()
    for badval in (255, -100, 3):
        lab = labels.clone()
        lab[1, 7, 9] = badval
        logits.grad = None
        v = fn(logits, lab)
        v.backward()
        assert math.isnan(v.item()) and bool(torch.isnan(logits.grad).any())
        with pytest.raises(mia_hip.MiaError):
            ops.check_labels()
        ops.check_labels()  # the check clears the verdict it has reported
    # sticky verdict: a CLEAN forward between the offending one and the check must not erase it
    lab = labels.clone()
    lab[0, 3, 3] = 7
    assert math.isnan(fn(logits, lab).item())
    assert math.isfinite(fn(logits, labels).item())  # e.g. a validation forward, a second loss term
    with pytest.raises(mia_hip.MiaError):
        ops.check_labels()
    ops.check_labels()
    with pytest.raises(NotImplementedError):
        fn2 = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs=dict(ignore_index=1))
        fn2.get_ce_loss(logits, labels)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c0,hw", [(16, (37, 53)), (32, (64, 64)), (64, (128, 80)), (96, (23, 100)), (128, (16, 16)), (8, (20, 28))])
def test_stem_forward(c0, hw, dtype):
    """Conv2d(1, C0, 3, padding=1) on the fp32 image (reference unet.py:54-66 with input_channels=1): the fp32 matrix-core
    kernel (C0 a multiple of 16; exact fp32 fma chain) and the VALU kernel (other C0) vs F.conv2d, outputs and the
    per-(image, channel) statistics partials; ragged widths (W not a multiple of 16), heights not a multiple of the bands."""
    import ctypes
    from mia_hip import call, lib, ops
    dev = _dev()
    g = torch.Generator().manual_seed(c0)
    n, (h, w) = 3, hw
    x = torch.rand(n, 1, h, w, generator=g)
    wt = torch.randn(c0, 1, 3, 3, generator=g) / 3
    b = torch.randn(c0, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    y = torch.empty((n, h, w, c0), device=dev, dtype=dtype)
    stats = torch.empty((n, lib().mia_stem_slabs(), c0, 2), device=dev, dtype=torch.float32)
    xd = x.to(dev).reshape(n, h, w, 1)
    wd_, bd = wt.reshape(c0, 9).to(dev).contiguous(), b.to(dev)  # named: a temporary would be freed (and its memory reused) before the launch
    call("mia_stem_fwd", ops._p(xd), ops._dt(xd), ops._p(wd_), ops._p(bd), ops._p(y), ops._dt(y), ops._p(stats), n, h, w, c0, ops._stream())
    got = nchw(y)
    assert relerr(got, ref) < (2e-6 if dtype == torch.float32 else 5e-3)
    s = stats.sum(1).cpu()
    assert relerr(s[..., 0], ref.sum((2, 3))) < 1e-4 and relerr(s[..., 1], (ref ** 2).sum((2, 3))) < 1e-4


def test_dropout_mask_zero_and_gather_utilities():
    """Library kernels that replace PyTorch's bernoulli / fill / strided copy inside the step: Dropout2d masks take the two
    values {0, 1/keep} with the right frequency, are reproducible under torch.manual_seed and differ call to call."""
    import ctypes
    from mia_hip import call, ops
    dev = _dev()
    torch.manual_seed(11)
    a = ops.dropout_mask(1 << 18, 0.9, dev)
    b = ops.dropout_mask(1 << 18, 0.9, dev)
    torch.manual_seed(11)
    a2 = ops.dropout_mask(1 << 18, 0.9, dev)
    assert torch.equal(a, a2) and not torch.equal(a, b)
    vals = torch.unique(a).cpu().tolist()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / 0.9) < 1e-6
    assert abs((a > 0).float().mean().item() - 0.9) < 5e-3 and abs(a.mean().item() - 1.0) < 5e-3
    odd = ops.dropout_mask(1001, 0.5, dev)  # tail of a quad
    assert odd.shape == (1001,) and abs((odd > 0).float().mean().item() - 0.5) < 0.06
    buf = torch.randn(4096 + 4, device=dev)
    call("mia_zero", ops._p(buf), ops._c_i64(4096 * 4), ops._stream())
    assert float(buf[:4096].abs().max()) == 0.0 and float(buf[4096:].abs().min()) > 0.0
    src = torch.arange(40, device=dev, dtype=torch.float32)
    dst = torch.empty(13, device=dev)
    call("mia_gather_f32", ops._p(src[1:]), ops._c_i64(3), ops._p(dst), 13, ops._stream())
    assert torch.equal(dst.cpu(), torch.arange(13, dtype=torch.float32) * 3 + 1)


@pytest.mark.parametrize("kind", ["adam", "adamw", "sgd"])
def test_optimizer_and_clip(kind):
    from mia_hip import ops, OPT_ADAM, OPT_ADAMW, OPT_SGD
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    nel = 10007
    p0 = torch.randn(nel, generator=g)
    pr = p0.clone().requires_grad_(True)
    if kind == "adam":
        opt = torch.optim.Adam([pr], betas=(0.9, 0.999), weight_decay=5e-4)
    elif kind == "adamw":
        opt = torch.optim.AdamW([pr], betas=(0.9, 0.999), weight_decay=5e-4)
    else:
        opt = torch.optim.SGD([pr], lr=1e-3, momentum=0.9, weight_decay=5e-4)
    pd = p0.to(dev)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    code = {"adam": OPT_ADAM, "adamw": OPT_ADAMW, "sgd": OPT_SGD}[kind]
    for step in range(1, 4):
        gr = torch.randn(nel, generator=g) * (3.0 if step == 2 else 0.01)
        pr.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 10.0)
        lr = 1e-3 * step
        for gp in opt.param_groups:
            gp["lr"] = lr
        opt.step()
        gd = gr.to(dev)
        clip = ops.grad_norm(gd, 10.0)
        ops.optim_step(code, pd, gd, m, v, lr, 0.9, 0.999, 1e-8, 5e-4, step, clip)
        torch.cuda.synchronize()
        assert abs(clip[0].item() - tn.item()) / tn.item() < 1e-5
        assert relerr(pd, pr) < 2e-6, (kind, step)


@pytest.mark.parametrize("k1,layout", [(3, "nchw"), (4, "nhwc"), (2, "nchw")])
def test_validation_argmax_dice_and_selector_scores(k1, layout):
    """argmax + per-class hard Dice (medpy.dc closed form) and the selectors' scores."""
    from activelearning.scores import selector_scores
    from metric.segmentation import predict_and_dice
    from oracle.losses_ref import hard_dice
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    b, h, w = 3, 40, 56
    logits = torch.randn(b, k1, h, w, generator=g) * 3
    labels = torch.randint(0, k1, (b, h, w), generator=g)
    labels[1][labels[1] == 1] = 0  # class 1 absent in image 1's ground truth
    logits[2, 1] = -50.0           # class 1 never predicted in image 2 -> Dice 0 by convention
    ld = logits.to(dev)
    if layout == "nhwc":
        ld = ld.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    pred, dice, counts = predict_and_dice(ld, labels.to(dev))
    ref_pred = logits.softmax(1).argmax(1)
    assert torch.equal(pred.cpu(), ref_pred)
    for i in range(b):
        for k in range(k1):
            assert abs(dice[i, k].item() - hard_dice(ref_pred[i] == k, labels[i] == k)) < 1e-6, (i, k)
    assert torch.equal(counts[..., 2].cpu().long(), torch.stack([(labels == k).sum((1, 2)) for k in range(k1)], 1))
    pred_only, d2, c2 = predict_and_dice(ld)
    assert torch.equal(pred_only.cpu(), ref_pred) and d2 is None and c2 is None
    # selector scores (entropy_selector.py:42-49, confidence_selector.py:42-47, margin_selector.py:42-48)
    prob = logits.softmax(1)
    ent = torch.mean(-prob * torch.log2(prob + 1e-8), dim=1).mean(dim=[-2, -1])
    conf = (-1 * prob.max(1)[0]).mean(dim=[-2, -1])
    top2 = prob.topk(2, dim=1)[0]
    marg = (-1 * (top2[:, 0] - top2[:, 1])).mean(dim=[-2, -1])
    sc = selector_scores(ld).cpu()
    np.testing.assert_allclose(sc[:, 0].numpy(), ent.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(sc[:, 1].numpy(), conf.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(sc[:, 2].numpy(), marg.numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("dtype,c", [(torch.float32, 64), (torch.bfloat16, 64), (torch.float32, 96), (torch.bfloat16, 32),
                                     (torch.float32, 24)])
def test_plain_block_two_output_gradients_are_summed_on_load(norm, dtype, c):
    """PlainBlockFn(dup=True) hands out its result twice (one storage); the two output gradients reach backward separately
    and are summed inside the norm kernels (c % 32 == 0) or by one explicit add (other widths).  Must equal a single
    backward with the pre-summed gradient."""
    from mia_hip import NORM_BATCH, NORM_INSTANCE, ops
    dev = _dev()
    g = torch.Generator().manual_seed(17)
    n, h, w, cin = 2, 24, 40, 16
    x = nhwc(q(torch.randn(n, cin, h, w, generator=g), dtype), dtype, dev)
    wt = (torch.randn(c, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev)
    b = torch.randn(c, generator=g).to(dev)
    ga = (1 + 0.1 * torch.randn(c, generator=g)).to(dev)
    be = (0.1 * torch.randn(c, generator=g)).to(dev)
    g1 = nhwc(q(torch.randn(n, c, h, w, generator=g), dtype), dtype, dev)
    g2 = nhwc(q(torch.randn(n, c, h, w, generator=g), dtype), dtype, dev)

    def cfg():
        if norm == "batch":
            return ops.NormCfg(NORM_BATCH, True, 1e-5, 0.1, torch.zeros(c, device=dev), torch.ones(c, device=dev),
                               torch.zeros((), device=dev, dtype=torch.long), None)
        return ops.NormCfg(NORM_INSTANCE, True, 1e-5, 0.1, None, None, None, None)

    def run(two):
        leaves = [t.clone().requires_grad_(True) for t in (x, wt, b, ga, be)]
        if two:
            za, zb = ops.PlainBlockFn.apply(leaves[0], None, leaves[1], leaves[2], leaves[3], leaves[4], 1, cfg(), None, 0.01, True)
            assert za.data_ptr() == zb.data_ptr()
            torch.autograd.backward([za, zb], [g1, g2])
        else:
            z = ops.PlainBlockFn.apply(leaves[0], None, leaves[1], leaves[2], leaves[3], leaves[4], 1, cfg())
            z.backward((g1.float() + g2.float()).to(dtype))
        torch.cuda.synchronize()
        return [t.grad.float() for t in leaves]

    got, want = run(True), run(False)
    tol = 1e-5 if dtype == torch.float32 else 2e-2  # bf16: the reference rounds g1 + g2 to bf16, the kernels sum in fp32
    for a_, b_ in zip(got, want):
        assert relerr(a_, b_) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["conv_xcd", "conv64", "conv64_dma", "wgrad_xcd", "wgrad_dma"])
def test_kernel_selection_knobs_do_not_change_results(knob):
    """mia_set_option knobs pick kernels / block orders only: the forward and input-gradient results are bit-identical either
    way (same per-element operation order), the weight gradient agrees to fp32 summation order."""
    import mia_hip
    from mia_hip import BF16, CONV_G3S1, CONV_T2S2, WGRAD_3S1, ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 40, 72, 64, generator=g).to(dev).to(torch.bfloat16)
    dy = torch.randn(2, 40, 72, 64, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
    wt = (torch.randn(64, 32, 2, 2, generator=g) / 16).to(dev)  # ConvTranspose2d(64 -> 32)
    b = torch.randn(64, generator=g).to(dev)
    bt = torch.randn(32, generator=g).to(dev)
    pc, pct = ops.PackCache(), ops.PackCache()
    wp, npad, kpad = pc.get(w, BF16, True)
    wtp, npt, kpt = pct.get(wt, BF16, False)
    lib = mia_hip.lib()

    def run():
        y, _, st = ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, 64, (40, 72), want_stats=True)
        up, _, _ = ops.conv_mma(CONV_T2S2, x, None, wtp, npt, kpt, False, bt, 32, (80, 144))
        dw = ops.conv_wgrad(WGRAD_3S1, x, None, dy, w.shape, 64, 64)
        return y.float().clone(), st.sum(1).clone(), up.float().clone(), dw.clone()

    default = mia_hip.get_option(knob)
    try:
        lib.mia_set_option(knob.encode(), 2 if knob == "conv64_dma" else 1)  # conv64_dma = 2: also the plain 64 -> 64 launches
        on = run()
        lib.mia_set_option(knob.encode(), 0)
        off = run()
    finally:
        lib.mia_set_option(knob.encode(), default)
    if knob not in ("conv64", "conv64_dma"):  # the 64-channel kernels accumulate taps in another order than the tile kernel / each other
        assert torch.equal(on[0], off[0])
    else:
        assert (on[0] - off[0]).abs().max().item() <= 2e-2 * off[0].abs().max().item()
    assert torch.allclose(on[1], off[1], rtol=1e-4, atol=1e-2)
    assert torch.equal(on[2], off[2])
    assert torch.allclose(on[3], off[3], rtol=1e-4, atol=1e-4 * off[3].abs().max().item())


PW_CASES = [  # n, cin, cout, h, w (coarse): ConvTranspose2d(cin -> cout, 2, 2) on conv_pw_kernel -- 64 outputs (a 128-column block
    # spans two taps), ragged last pixel tile, pixel tiles not a multiple of 8 (column-block-slow item order), odd widths,
    # more items than CUs, K = 128 (two stages per item) and K = 1024
    (2, 128, 64, 64, 64), (1, 128, 64, 7, 9), (3, 256, 128, 20, 13), (1, 1024, 512, 8, 8), (1, 512, 256, 16, 16), (4, 128, 64, 128, 96),
    (1, 192, 96, 24, 24),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", PW_CASES)
def test_conv_pw_matches_tile_kernel_and_reference(case):
    """Option conv_pw (csrc/conv_pw.hip): the forward accumulates K in the same order as the tile kernel -> bit-identical; the
    input gradient sums (tap, channel) in another order -> fp32 rounding only.  Both against F.conv_transpose2d on the
    bf16-quantised operands (reference: unet.py:142 / :212)."""
    import mia_hip
    from mia_hip import BF16, CONV_G2S2, CONV_T2S2, ops
    dev = _dev()
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    dt = torch.bfloat16
    x = q(torch.randn(n, cin, h, w, generator=g), dt)
    wt = torch.randn(cin, cout, 2, 2, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    dy = q(torch.randn(n, cout, 2 * h, 2 * w, generator=g), dt)
    xr = x.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, q(wt, dt), b, stride=2)
    yr.backward(dy)
    xd, dyd, wd = nhwc(x, dt, dev), nhwc(dy, dt, dev), wt.to(dev)
    pc = ops.PackCache()
    wf, nf, kf = pc.get(wd, BF16, False)
    wb, nb, kb = pc.get(wd, BF16, True)
    outs = {}
    try:
        for flag in (1, 0):
            mia_hip.set_option("conv_pw", flag)
            y, _, _ = ops.conv_mma(CONV_T2S2, xd, None, wf, nf, kf, False, b.to(dev), cout, (2 * h, 2 * w))
            dx, _, _ = ops.conv_mma(CONV_G2S2, dyd, None, wb, nb, kb, False, None, cin, (h, w))
            torch.cuda.synchronize()
            outs[flag] = (y.clone(), dx.clone())
    finally:
        mia_hip.set_option("conv_pw", 1)
    assert torch.equal(outs[1][0], outs[0][0])
    assert relerr(outs[1][1].float(), outs[0][1].float()) < 1e-2  # one bf16 ulp (2^-8) where the fp32 sums round apart
    assert relerr(nchw(outs[1][0]), yr) < TOL[dt]
    assert relerr(nchw(outs[1][1]), xr.grad) < TOL[dt]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 128, 64, 64, 64), (1, 256, 128, 17, 23), (3, 128, 64, 9, 40), (1, 1024, 512, 8, 8), (1, 384, 192, 20, 12)])
def test_wgrad_t2_matches_fast_kernel_and_reference(case):
    """Option wgrad_t2 (csrc/conv_wgrad.hip::wgrad_bf16_bt_t2_kernel): the ConvTranspose2d 2x2 weight gradient on the 512-thread
    three-stage ring; fp32 slabs of exact bf16 products -> agreement with the 256-thread kernel and with autograd to fp32 summation
    order (reference: unet.py:142)."""
    import mia_hip
    from mia_hip import WGRAD_2S2, ops
    dev = _dev()
    n, cin, cout, h, w = case  # ConvTranspose2d(cin -> cout): coarse h x w, fine 2h x 2w
    g = torch.Generator().manual_seed(sum(case))
    dt = torch.bfloat16
    x = q(torch.randn(n, cin, h, w, generator=g), dt)
    wt = (torch.randn(cin, cout, 2, 2, generator=g) / math.sqrt(cin)).requires_grad_(True)
    dout = q(torch.randn(n, cout, 2 * h, 2 * w, generator=g), dt)
    F.conv_transpose2d(x, wt, None, stride=2).backward(dout)
    xd, dd = nhwc(x, dt, dev), nhwc(dout, dt, dev)
    outs = {}
    try:
        for flag in (1, 0):
            mia_hip.set_option("wgrad_t2", flag)
            outs[flag] = ops.conv_wgrad(WGRAD_2S2, dd, None, xd, wt.shape, cin, cout).clone()
            torch.cuda.synchronize()
    finally:
        mia_hip.set_option("wgrad_t2", 1)
    assert relerr(outs[1], wt.grad) < 1e-4 and relerr(outs[0], wt.grad) < 1e-4
    assert torch.allclose(outs[1], outs[0], rtol=1e-4, atol=1e-4 * outs[0].abs().max().item())


PW_S2_CASES = [  # n, cin, cout, h, w (input): strided 3x3 forward on conv_pw_kernel<G3S2> -- image borders on all four sides, odd input
    # sizes (the last row / column tap falls outside), K = 9 x 64 .. 9 x 512, with and without the statistics epilogue
    (2, 64, 128, 64, 64), (1, 128, 256, 32, 64), (3, 64, 128, 31, 63), (1, 512, 1024, 16, 32), (2, 128, 128, 128, 32),
]


@pytest.mark.gpu
@pytest.mark.parametrize("want_stats", [True, False])
@pytest.mark.parametrize("case", PW_S2_CASES)
def test_conv_pw_s2_matches_tile_kernel_and_reference(case, want_stats):
    """Option conv_pw_s2 (csrc/conv_pw.hip, MODE_G3S2): the strided first conv of an encoder level (unet.py:57-66) as a tap-gathered
    GEMM.  K is summed tap-major (the tile kernel: chunk-major) -> fp32 order only.  Statistics: another partition of the image's
    pixels into the same number of entries; their per-image sums are what the norm consumes."""
    import mia_hip
    from mia_hip import BF16, CONV_G3S2, ops
    dev = _dev()
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    dt = torch.bfloat16
    x = q(torch.randn(n, cin, h, w, generator=g), dt)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    yr = F.conv2d(x, q(wt, dt), b, stride=2, padding=1)
    xd = nhwc(x, dt, dev)
    wp, npad, kpad = ops.PackCache().get(wt.to(dev), BF16, True)
    outs = {}
    try:
        for flag in (1, 0):
            mia_hip.set_option("conv_pw_s2", 2 * flag)  # 2: also above 256 input channels
            y, _, st = ops.conv_mma(CONV_G3S2, xd, None, wp, npad, kpad, False, b.to(dev), cout, (ho, wo), want_stats=want_stats)
            torch.cuda.synchronize()
            outs[flag] = (y.clone(), None if st is None else st.clone())
    finally:
        mia_hip.set_option("conv_pw_s2", 0)
    assert relerr(outs[1][0].float(), outs[0][0].float()) < 1e-2  # one bf16 ulp where the fp32 sums round apart
    assert relerr(nchw(outs[1][0]), yr) < TOL[dt]
    if want_stats:
        assert outs[1][1].shape == outs[0][1].shape
        s1, s0 = outs[1][1].sum(1).cpu(), outs[0][1].sum(1).cpu()
        assert torch.allclose(s1, s0, rtol=1e-3, atol=1e-1)
        assert relerr(s1[..., 0], yr.sum((2, 3))) < 1e-3 + TOL[dt]
        assert relerr(s1[..., 1], (yr ** 2).sum((2, 3))) < 1e-3 + TOL[dt]


S2_WIDE_CASES = [  # n, cin, cout, h, w: 512-thread stride-2 kernel (128-multiples of output channels), ragged 16-row tiles, odd sizes
    (2, 64, 128, 80, 72), (1, 128, 256, 37, 53), (1, 32, 128, 18, 200), (3, 64, 128, 34, 34), (1, 256, 512, 32, 32),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", S2_WIDE_CASES)
def test_conv_s2_wide_matches_tile_kernel_and_reference(case):
    """Option conv_s2_wide (csrc/conv_mma_fast.hip, WC = 2): same chunk / tap order per output element as the 256-thread tile
    kernel -> bit-identical outputs; the 8-row statistics tiles are summed from other partials (fp32 order).  Both against
    F.conv2d on the bf16-quantised operands (reference: unet.py:57-66, the strided first block of an encoder level)."""
    import mia_hip
    from mia_hip import BF16, CONV_G3S2, ops
    dev = _dev()
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    x = q(torch.randn(n, cin, h, w, generator=g), torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    yr = F.conv2d(x, q(wt, torch.bfloat16), b, stride=2, padding=1)
    xd = nhwc(x, torch.bfloat16, dev)
    wp, npad, kpad = ops.PackCache().get(wt.to(dev), BF16, True)
    outs = {}
    try:
        mia_hip.set_option("conv_pw_s2", 0)  # (the tap-gathered GEMM would take these launches)
        for flag in (1, 0):
            mia_hip.set_option("conv_s2_wide", 2 * flag)  # 2: also below 128 input channels
            y, _, st = ops.conv_mma(CONV_G3S2, xd, None, wp, npad, kpad, False, b.to(dev), cout, (ho, wo), want_stats=True)
            torch.cuda.synchronize()
            outs[flag] = (y.clone(), st.clone())
    finally:
        mia_hip.set_option("conv_s2_wide", 1)
        mia_hip.set_option("conv_pw_s2", 0)
    assert torch.equal(outs[1][0], outs[0][0])
    assert outs[1][1].shape == outs[0][1].shape
    assert torch.allclose(outs[1][1], outs[0][1], rtol=1e-4, atol=1e-2)
    assert relerr(nchw(outs[1][0]), yr) < TOL[torch.bfloat16]
    s = outs[1][1].sum(1).cpu()
    assert relerr(s[..., 0], yr.sum((2, 3))) < 1e-3 + TOL[torch.bfloat16]
    assert relerr(s[..., 1], (yr ** 2).sum((2, 3))) < 1e-3 + TOL[torch.bfloat16]


BT_CASES = [
    # n, c1, c2, cout, h, w  -- conv_bt.hip: 128 / 96 / 64-channel blocks, ragged 16 x 32 tiles, two sources, two destinations,
    # one tile per workgroup and several (persistent work list: more items than CUs), maps of one tile row
    (2, 128, 0, 128, 40, 72), (1, 256, 0, 128, 17, 33), (1, 128, 128, 128, 48, 48), (1, 96, 0, 192, 24, 40),
    (1, 96, 96, 96, 33, 70), (2, 64, 64, 64, 40, 56), (3, 128, 0, 256, 96, 160), (1, 512, 0, 512, 9, 31),
    (1, 64, 0, 128, 16, 32), (1, 160, 0, 384, 20, 20),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", BT_CASES)
def test_conv_bt_matches_tile_kernel_and_reference(case):
    """The big-tile LDS-DMA kernel (csrc/conv_bt.hip; option conv_bt) against the tile kernel it replaces and against fp32-CPU
    F.conv2d on shared bf16 operands: forward (+ epilogue statistics) and input gradient (two destinations for two sources).
    The two kernels differ only by where the bias enters the fp32 sum (first vs last) and the final bf16 rounding."""
    import mia_hip
    from mia_hip import BF16, CONV_G3S1, ops
    dev = _dev()
    n, c1, c2, cout, h, w = case
    cin = c1 + c2
    g = torch.Generator().manual_seed(c1 * 7 + cout + h)
    x = q(torch.randn(n, cin, h, w, generator=g), torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    dy = q(torch.randn(n, cout, h, w, generator=g), torch.bfloat16)
    xr, wr = x.clone().requires_grad_(True), q(wt, torch.bfloat16).clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=1)
    yr.backward(dy)
    x1 = nhwc(x[:, :c1], torch.bfloat16, dev)
    x2 = nhwc(x[:, c1:], torch.bfloat16, dev) if c2 else None
    dyd = nhwc(dy, torch.bfloat16, dev)
    pc = ops.PackCache()
    wp, npad, kpad = pc.get(wt.to(dev), BF16, True)
    wb, npb, kpb = pc.get(wt.to(dev), BF16, False)

    def run():
        y, _, st = ops.conv_mma(CONV_G3S1, x1, x2, wp, npad, kpad, False, b.to(dev), cout, (h, w), want_stats=True)
        dx1, dx2, dst = ops.conv_mma(CONV_G3S1, dyd, None, wb, npb, kpb, True, None, cin, (h, w), want_stats=True,
                                     out_split=c1 if c2 else None)
        dx = nchw(dx1) if dx2 is None else torch.cat([nchw(dx1), nchw(dx2)], 1)
        return nchw(y), st.sum(1).cpu(), dx, dst.sum(1).cpu()

    try:
        mia_hip.set_option("conv_bt", 1)
        on = run()
        mia_hip.set_option("conv_bt", 0)
        off = run()
    finally:
        mia_hip.set_option("conv_bt", 1)
    assert relerr(on[0], yr) < 6e-3 and relerr(on[2], xr.grad) < 6e-3
    assert relerr(on[1][..., 0], yr.detach().sum((2, 3))) < 1e-3 and relerr(on[1][..., 1], (yr.detach() ** 2).sum((2, 3))) < 1e-3
    assert relerr(on[3][..., 0], xr.grad.sum((2, 3))) < 2e-3
    # kernel vs kernel: at most one bf16 rounding step apart (2^-8 relative per element), statistics to fp32 summation order
    assert (on[0] - off[0]).abs().max().item() <= 2 ** -7 * off[0].abs().max().item()
    assert (on[2] - off[2]).abs().max().item() <= 2 ** -7 * off[2].abs().max().item()
    assert torch.allclose(on[1], off[1], rtol=1e-4, atol=1e-3 * off[1].abs().max().item())
    # weight gradient: the 512-thread 128 x 64 block kernel (option wgrad_bt; >= 128 output channels) against the ring kernel
    # it replaces and against the fp32 reference (fp32 slabs of exact bf16 products: summation order only)
    from mia_hip import WGRAD_3S1
    try:
        mia_hip.set_option("wgrad_bt", 1)
        dw_on = ops.conv_wgrad(WGRAD_3S1, x1, x2, dyd, wt.shape, cout, cin)
        mia_hip.set_option("wgrad_bt", 0)
        dw_off = ops.conv_wgrad(WGRAD_3S1, x1, x2, dyd, wt.shape, cout, cin)
    finally:
        mia_hip.set_option("wgrad_bt", 1)
    assert relerr(dw_on, wr.grad) < 1e-4 and relerr(dw_off, wr.grad) < 1e-4
    assert torch.allclose(dw_on, dw_off, rtol=1e-4, atol=1e-4 * dw_off.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("pieces", [1, 2])
@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("c,hw", [(256, (24, 40)), (1024, (8, 12)), (3072, (3, 5)), (96, (48, 80)), (1536, (6, 6))])
def test_norm_streams_wide_bf16_vs_fp32_cpu(c, hw, norm, pieces):
    """The normalisation streams at the BENCHMARKED widths in bf16 (20 % of the step, previously pinned only
    at C <= 96): mia_norm_finalize -> mia_norm_act_fwd -> mia_norm_act_bwd (column reduce + finalize + apply; output gradient
    in one or two pieces summed on load) against fp32 CPU autograd of InstanceNorm2d / BatchNorm2d(train) + LeakyReLU(0.01)
    (blocks.py:98-102) on the SAME bf16-rounded y, dz.  Bounds: bf16 outputs one rounding step (2^-8 of the tensor max),
    fp32 parameter gradients 1e-3 relative."""
    from mia_hip import BF16, NORM_BATCH, NORM_INSTANCE, call, lib, ops
    from mia_hip.ops import _c_float, _c_i64, _p, _stream
    dev = _dev()
    h, w = hw
    n = 2
    g = torch.Generator().manual_seed(c + h + pieces)
    y = q(torch.randn(n, c, h, w, generator=g) * (1 + torch.rand(1, c, 1, 1, generator=g)) + torch.randn(1, c, 1, 1, generator=g),
          torch.bfloat16)
    gamma = 1 + 0.2 * torch.randn(c, generator=g)
    beta = 0.2 * torch.randn(c, generator=g)
    dzs = [q(torch.randn(n, c, h, w, generator=g), torch.bfloat16) for _ in range(pieces)]
    # ---- fp32 CPU reference
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    if norm == "instance":
        zn = F.instance_norm(yr, weight=gr, bias=br, eps=1e-5)
    else:
        zn = F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5)
    zr = F.leaky_relu(zn, 0.01)
    zr.backward(sum(dzs))
    # ---- kernels
    yd = nhwc(y, torch.bfloat16, dev)
    dzd = [nhwc(t, torch.bfloat16, dev) for t in dzs]
    if not lib().mia_norm_two_piece_ok(BF16, c) and pieces == 2:
        pytest.skip("two-piece path not built for this width")
    yf = yd.float().reshape(n, h * w, c)
    stats = torch.stack([yf.sum(1), (yf * yf).sum(1)], -1).reshape(n, 1, c, 2).contiguous()  # what the conv epilogue delivers
    mode = NORM_INSTANCE if norm == "instance" else NORM_BATCH
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), device=dev, dtype=torch.long)
    coefs = torch.empty((5, n, c), device=dev, dtype=torch.float32)
    gd, bd = gamma.to(dev), beta.to(dev)
    call("mia_norm_finalize", _p(stats), n, 1, c, _c_i64(h * w), mode, 1, None, _p(gd), _p(bd), _c_float(1e-5), _c_float(0.1),
         _p(rm) if norm == "batch" else None, _p(rv) if norm == "batch" else None, _p(nbt) if norm == "batch" else None,
         _p(coefs[0]), _p(coefs[1]), _p(coefs[2]), _p(coefs[3]), _p(coefs[4]), _stream())
    z = torch.empty_like(yd)
    call("mia_norm_act_fwd", _p(yd), _p(z), BF16, _p(coefs[2]), _p(coefs[3]), n, _c_i64(h * w), c, _c_float(0.01), None, _stream())
    slabs = ops._slabs_for(h * w)
    part = torch.empty((n, slabs, c, 2), device=dev, dtype=torch.float32)
    cc = torch.empty((2, n, c), device=dev, dtype=torch.float32)
    dgb = torch.empty((3, c), device=dev, dtype=torch.float32)
    dy = torch.empty_like(yd)
    call("mia_norm_act_bwd", _p(dzd[0]), _p(dzd[1]) if pieces == 2 else None, _p(yd), _p(dy), BF16, _p(coefs[2]), _p(coefs[3]),
         _p(coefs[0]), _p(coefs[1]), _p(coefs[4]), n, _c_i64(h * w), c, mode, 0, _c_float(0.01), slabs, _p(part), _p(cc[0]), _p(cc[1]),
         _p(dgb[0]), _p(dgb[1]), _p(dgb[2]), 0, None, _stream())
    torch.cuda.synchronize()
    step = 2.0 ** -8
    assert relerr(nchw(z), zr) < 1.5 * step
    assert relerr(nchw(dy), yr.grad) < 2.5 * step  # bf16 output + the rounding of the recomputed activation sign / xhat
    assert relerr(dgb[0], gr.grad) < 1e-3 and relerr(dgb[1], br.grad) < 1e-3
    # conv-bias gradient = sum of dy over (n, h, w) (closed form from the reduction sums, norm_bwd_finalize): ~0 analytically for
    # these norms (the mean is removed), so compare absolutely against the scale of |dy| sums
    ref_db = yr.grad.sum((0, 2, 3))
    assert (dgb[2].cpu() - ref_db).abs().max().item() < 1e-3 * yr.grad.abs().sum((0, 2, 3)).max().item() + 1e-4
    if norm == "batch":
        np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * y.mean((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * y.var((0, 2, 3), unbiased=True)).numpy(), rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 64, 128, 40, 72), (1, 128, 256, 17, 33), (1, 256, 512, 64, 64), (3, 96, 384, 23, 41),
                                  (1, 512, 1024, 9, 16), (1, 64, 128, 130, 260)])
def test_wgrad_bt_stride2_matches_fast_kernel_and_reference(case):
    """Weight gradient of the stride-2 3x3 conv (first block of an encoder level, unet.py:57) on the 512-thread LDS-DMA kernel
    (wgrad_bf16_bt_s2_kernel; option wgrad_bt) against the register-staged kernel it replaces and fp32-CPU autograd on shared bf16
    operands: ragged tiles, odd sizes, several split-K rounds, more tiles than workgroups."""
    import mia_hip
    from mia_hip import WGRAD_3S2, ops
    dev = _dev()
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(cin + h)
    x = q(torch.randn(n, cin, h, w, generator=g), torch.bfloat16)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    dy = q(torch.randn(n, cout, ho, wo, generator=g), torch.bfloat16)
    wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(x, wr, None, stride=2, padding=1).backward(dy)
    xd, dyd = nhwc(x, torch.bfloat16, dev), nhwc(dy, torch.bfloat16, dev)
    try:
        mia_hip.set_option("wgrad_bt", 1)
        on = ops.conv_wgrad(WGRAD_3S2, xd, None, dyd, wr.shape, cout, cin)
        mia_hip.set_option("wgrad_bt", 0)
        off = ops.conv_wgrad(WGRAD_3S2, xd, None, dyd, wr.shape, cout, cin)
    finally:
        mia_hip.set_option("wgrad_bt", 1)
    assert relerr(on, wr.grad) < 1e-4 and relerr(off, wr.grad) < 1e-4
    assert torch.allclose(on, off, rtol=1e-4, atol=1e-4 * off.abs().max().item())


# ---------------------------------------------------------------- fused PlainBlock: normalise-on-load (round 4)
NL_CASES = [(2, 37, 53), (1, 16, 16), (3, 64, 48), (2, 128, 128), (1, 9, 200), (5, 33, 17)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", NL_CASES)
def test_conv_and_wgrad_normalise_on_load_match_materialised_path(case):
    """`mia_conv_mma_nl` / `mia_conv_wgrad_nl` (the consumer half of the fused PlainBlock, blocks.py:98-102 folded into the next
    conv) against the two-pass path they replace -- `mia_norm_act_fwd` then `mia_conv_mma` / `mia_conv_wgrad` -- on the same
    raw producer output and coefficient table.  The forward must be BIT-IDENTICAL (same fp32 fma / select / RNE, same conv
    kernel); the weight gradient is bit-identical to the register-staged kernel it is built on (option wgrad_dma = 0) and
    within fp32 summation order of the default LDS-DMA kernel.  Ragged tiles, border tiles, several images, image count not
    a divisor of the tile walk; per-(n, c) coefficients of both signs incl. zero scale (Dropout2d-dropped channels)."""
    import mia_hip
    from mia_hip import CONV_G3S1, WGRAD_3S1, call, ops
    from mia_hip.ops import _c_float, _c_i64, _p, _stream
    dev = _dev()
    n, h, w = case
    c = 64
    g = torch.Generator().manual_seed(100 + n * h + w)
    y = (torch.randn(n, h, w, c, generator=g) * 1.5).to(dev, torch.bfloat16)
    coefs = torch.zeros(5, n, c)
    coefs[2] = torch.randn(n, c, generator=g)            # scale (both signs)
    coefs[3] = torch.randn(n, c, generator=g) * 0.7      # shift
    coefs[2][:, 5] = 0.0                                 # a dropped channel: z = lrelu(shift)
    coefs[2][0, 9], coefs[3][0, 9] = 0.0, 0.0
    coefs = coefs.to(dev)
    wt = (torch.randn(c, c, 3, 3, generator=g) / 24).to(dev)
    bias = torch.randn(c, generator=g).to(dev)
    dy = torch.randn(n, h, w, c, generator=g).to(dev, torch.bfloat16)
    slope = 0.01
    pc = ops.PackCache()
    wp, npad, kpad = pc.get(wt, mia_hip.BF16, True)
    z = torch.empty_like(y)
    call("mia_norm_act_fwd", _p(y), _p(z), mia_hip.BF16, _p(coefs[2]), _p(coefs[3]), n, _c_i64(h * w), c, _c_float(slope), None, _stream())
    ref, _, ref_stats = ops.conv_mma(CONV_G3S1, z, None, wp, npad, kpad, False, bias, c, (h, w), want_stats=True)
    if not ops.nl_supported(torch.bfloat16, c, c, h, w, True):
        assert h <= 8
        return
    got, _, got_stats = ops.conv_mma(CONV_G3S1, y, None, wp, npad, kpad, False, bias, c, (h, w), want_stats=True, nl=(coefs, slope))
    assert torch.equal(got, ref)
    assert torch.equal(got_stats, ref_stats)
    # torch-CPU fp32 math on the same bf16 operands
    zc = z.float().cpu().permute(0, 3, 1, 2)
    want = F.conv2d(zc, wt.cpu().to(torch.bfloat16).float(), bias.cpu(), padding=1)
    assert relerr(nchw(got), want) < 1e-2
    # weight gradient (the split-K count pinned to the plain rule: the cost model of ops._ksplit_by_cost looks at the tile count, which differs
    # between the kernels compared below, and bit-identity needs one summation order)
    ops.WGRAD_KSPLIT_MODEL = False
    dw_ref = ops.conv_wgrad(WGRAD_3S1, z, None, dy, wt.shape, c, c)
    dw_nl = ops.conv_wgrad(WGRAD_3S1, y, None, dy, wt.shape, c, c, nl=(coefs, slope))
    want_dw = torch.nn.grad.conv2d_weight(zc, wt.shape, dy.float().cpu().permute(0, 3, 1, 2), padding=1)
    assert relerr(dw_nl, want_dw) < 2e-4 and relerr(dw_ref, want_dw) < 2e-4
    assert relerr(dw_nl, dw_ref) < 2e-5
    old = mia_hip.get_option("wgrad_dma")
    try:
        mia_hip.set_option("wgrad_dma", 0)
        dw_2wg = ops.conv_wgrad(WGRAD_3S1, z, None, dy, wt.shape, c, c)
    finally:
        mia_hip.set_option("wgrad_dma", old)
        ops.WGRAD_KSPLIT_MODEL = True
    assert torch.equal(dw_nl, dw_2wg)


@pytest.mark.gpu
@pytest.mark.parametrize("norm,drop", [("instance", None), ("batch", None), ("instance", 0.3)])
def test_fused_level_pairs_match_unfused_model(norm, drop):
    """A bf16 UNet whose level-0 blocks are 64 channels wide runs stem -> encoder.levels.0.1 and decoder.levels.-1.0 ->
    decoder.levels.-1.1 (+ head) as fused pairs (ops.LazyAct).  Against the same model with `ops.FUSE_NL = False`: logits
    and every input-side gradient bit-identical, the two consumer blocks' weight gradients within fp32 summation order (a
    different weight-gradient kernel serves them), in train and eval mode, with Dropout2d masks and batch norm."""
    from losses.compound_losses import DiceAndCELoss
    from mia_hip import ops
    from models.unet import UNet
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 1, 48, 80, generator=g).to(dev)
    lab = torch.randint(0, 3, (3, 48, 80), generator=g).to(dev)
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    res = {}
    old, old_cr = ops.FUSE_NL, ops.FUSE_CR
    ops.FUSE_CR = False  # (the backward reduction fused into the input-gradient epilogue has its own test; it is not bit-identical)
    try:
        for fuse in (False, True):
            ops.FUSE_NL = fuse
            torch.manual_seed(11)
            m = UNet(2, 1, 3, [64, 128], normalization=norm, dropout_prob=drop).to(dev)
            m.set_compute_dtype(torch.bfloat16)
            gp = torch.Generator().manual_seed(17)  # the same perturbation of the norm affine vectors / biases in both runs
            with torch.no_grad():
                for p in m.parameters():
                    if p.ndim == 1:
                        p.add_(0.1 * torch.randn(p.shape, generator=gp).to(dev))
            m.train()
            torch.manual_seed(5)  # Dropout2d masks come from the device generator
            torch.cuda.manual_seed(5)
            out = m(x)
            loss = loss_fn(out, lab)
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
            m.eval()
            with torch.no_grad():
                ev = m(x).clone()
                feat = m.get_pixel_feature(x)[1].clone()
            res[fuse] = (out.detach().clone(), loss.item(), grads, ev, feat)
    finally:
        ops.FUSE_NL, ops.FUSE_CR = old, old_cr
    a, b = res[False], res[True]
    assert torch.equal(a[0], b[0]) and a[1] == b[1]
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    consumers = ("encoder.levels.0.1.all.0.weight", "decoder.levels.0.1.all.0.weight")
    for k in a[2]:
        if k in consumers:
            assert relerr(b[2][k], a[2][k]) < 5e-5, k
        else:
            assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.gpu
def test_reserve_cus_keeps_results(monkeypatch):
    """Library option `reserve_cus` (room for RCCL's ring kernels under data parallelism; include/mia_hip.h): the persistent
    grids of conv_bt / conv_pw / conv64 / conv64_dma shrink to CUs - k workgroups over the SAME work items -- logits and every
    activation gradient bit-identical for k in {0, 8, 24}; the weight gradients' split-K count follows k (another fixed fp32
    summation order): equal within 1e-5 of the tensor's max, and run-to-run identical at a given k."""
    import mia_hip
    from losses.compound_losses import DiceAndCELoss
    from models.unet import UNet
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, 1, 128, 160, generator=g).to(dev)
    lab = torch.randint(0, 3, (2, 128, 160), generator=g).to(dev)
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    torch.manual_seed(21)
    m = UNet(2, 1, 3, [64, 128, 256], normalization="instance", dropout_prob=None).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    old = mia_hip.get_option("reserve_cus")
    res = {}
    try:
        for k in (0, 8, 24, 8):
            mia_hip.set_option("reserve_cus", k)
            m.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            out = m(xin)
            loss_fn(out, lab).backward()
            rec = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
            if k in res:  # second run at k = 8: bit-identical to the first
                assert torch.equal(rec[0], res[k][0])
                for n in rec[1]:
                    assert torch.equal(rec[1][n], res[k][1][n]), n
            res[k] = rec
    finally:
        mia_hip.set_option("reserve_cus", old)
    for k in (8, 24):
        assert torch.equal(res[k][0], res[0][0])
        for n, gk in res[k][1].items():
            g0 = res[0][1][n]
            if n.endswith("all.0.weight") or "upsamples" in n and n.endswith("weight"):
                assert relerr(gk, g0) < 1e-5, (k, n)
            else:  # biases / norm affine / head: no split-K dependence
                assert torch.equal(gk, g0), (k, n)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("norm,c0,hw", [("instance", 64, (48, 80)), ("batch", 16, (38, 54)), ("instance", 32, (130, 70))])
def test_stem_backward_fused_into_weight_gradient(norm, c0, hw, dtype):
    """The stem block has no input gradient, so its weight gradient is the only consumer of the conv-output gradient dy:
    `mia_norm_bwd_sums` + `mia_stem_wgrad_fused` (dy formed on load from dz, y and the coefficient rows) against the path it
    replaces (`mia_norm_act_bwd` writing dy, `mia_stem_wgrad` reading it): every gradient of the model BIT-IDENTICAL (the two
    paths share one definition of the element-wise arithmetic, csrc/common.h::norm_bwd_dy, and round dy to the storage dtype
    at the same point), and the stem's weight gradient against fp32-CPU autograd of the reference block."""
    from mia_hip import ops
    from models.unet import UNet
    dev = _dev()
    g = torch.Generator().manual_seed(c0 + hw[0])
    x = torch.rand(3, 1, *hw, generator=g).to(dev)
    gz = torch.randn(3, 3, *hw, generator=g).to(dev)
    res = {}
    old = ops.FUSE_STEM_BWD
    try:
        for fuse in (False, True):
            ops.FUSE_STEM_BWD = fuse
            torch.manual_seed(4)
            m = UNet(2, 1, 3, [c0, 2 * c0], normalization=norm, dropout_prob=None).to(dev)
            m.set_compute_dtype(dtype)
            m.train()
            (m(x) * gz).sum().backward()
            res[fuse] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    finally:
        ops.FUSE_STEM_BWD = old
    for k in res[False]:
        assert torch.equal(res[False][k], res[True][k]), k
    # block-level check of the fused path against torch-CPU autograd (fp32 only: exact operands)
    if dtype == torch.float32:
        blk = torch.nn.Sequential(torch.nn.Conv2d(1, c0, 3, padding=1),
                                  (torch.nn.InstanceNorm2d if norm == "instance" else torch.nn.BatchNorm2d)(c0, eps=1e-5, affine=True),
                                  torch.nn.LeakyReLU(0.01))
        sd = m.encoder.levels[0][0].all.state_dict()
        blk.load_state_dict({"0.weight": sd["0.weight"].cpu(), "0.bias": sd["0.bias"].cpu(), "1.weight": sd["2.weight"].cpu(),
                             "1.bias": sd["2.bias"].cpu(), **({"1.running_mean": torch.zeros(c0), "1.running_var": torch.ones(c0),
                                                               "1.num_batches_tracked": torch.tensor(0)} if norm == "batch" else {})})
        from models.unet.blocks import PlainBlock
        pb = PlainBlock(2, 1, c0, normalization=norm).to(dev)
        pb.all.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        pb.train()
        gg = torch.randn(3, c0, *hw, generator=g)
        ops.FUSE_STEM_BWD = True
        try:
            (pb(x) * gg.to(dev)).sum().backward()
        finally:
            ops.FUSE_STEM_BWD = old
        (blk.train()(x.cpu()) * gg).sum().backward()
        assert relerr(pb.all[0].weight.grad, blk[0].weight.grad) < 2e-4
        assert relerr(pb.all[2].weight.grad, blk[1].weight.grad) < 2e-4
        assert relerr(pb.all[2].bias.grad, blk[1].bias.grad) < 2e-4


@_needs_experiments
@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 37, 53), (1, 16, 16), (3, 64, 48), (5, 33, 17), (32, 128, 128), (2, 512, 384)])
def test_conv_input_gradient_with_column_reduce_epilogue(case):
    """`mia_conv_mma_cr`: the input-gradient conv of the consuming block with the producing block's norm-backward reduction in
    its epilogue.  dz bit-identical to `mia_conv_mma`; the per-tile partial sums, added up, equal sum g and sum g * xhat
    (g = dz * lrelu'(scale * y + shift)) computed in fp64 from the stored bf16 dz -- ragged tiles, several images, and more tiles
    than workgroups (the persistent walk with the next tile's prefetch live across the epilogue) included;
    `mia_norm_act_bwd_pre` on them gives the dy / dgamma / dbeta of `mia_norm_act_bwd` within fp32 summation order."""
    import mia_hip
    from mia_hip import CONV_G3S1, NORM_INSTANCE, call, ops
    from mia_hip.ops import _c_float, _c_i64, _p, _stream
    dev = _dev()
    n, h, w = case
    c = 64
    g = torch.Generator().manual_seed(7 * n + h)
    dyb = torch.randn(n, h, w, c, generator=g).to(dev, torch.bfloat16)
    y = (torch.randn(n, h, w, c, generator=g) * 1.5).to(dev, torch.bfloat16)
    coefs = torch.zeros(5, n, c)
    coefs[0] = torch.rand(n, c, generator=g) + 0.5     # xa
    coefs[1] = torch.randn(n, c, generator=g) * 0.3    # xb
    coefs[2] = torch.randn(n, c, generator=g)          # scale
    coefs[3] = torch.randn(n, c, generator=g) * 0.7    # shift
    coefs = coefs.to(dev)
    wt = (torch.randn(c, c, 3, 3, generator=g) / 24).to(dev)
    pc = ops.PackCache()
    wb, npad, kpad = pc.get(wt, mia_hip.BF16, False)
    ref, _, _ = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w))
    got, _, part = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w), cr=(y, coefs, 0.01))
    assert torch.equal(got, ref)
    dz, yf = got.double().cpu(), y.double().cpu()
    cf = coefs.double().cpu()
    u = cf[2][:, None, None, :] * yf + cf[3][:, None, None, :]
    gg = torch.where(u > 0, dz, dz * 0.01)
    xhat = cf[0][:, None, None, :] * yf + cf[1][:, None, None, :]
    want1, want2 = gg.sum((1, 2)), (gg * xhat).sum((1, 2))
    p = part.double().cpu().sum(1)
    scale_ref = max(float(want1.abs().max()), float(want2.abs().max()))
    assert float((p[..., 0] - want1).abs().max()) < 2e-5 * scale_ref + 1e-3
    assert float((p[..., 1] - want2).abs().max()) < 2e-5 * scale_ref + 1e-3
    # the block's norm backward from the partials vs its own reduction pass
    gam = torch.ones(c, device=dev)
    outs = []
    for use_pre in (False, True):
        dy = torch.empty_like(y)
        cc = torch.empty(2, n, c, device=dev)
        dgb = torch.empty(3, c, device=dev)
        slabs = max(1, min(64, h * w // 1024))
        pt = torch.empty(n, slabs, c, 2, device=dev)
        if use_pre:
            call("mia_norm_act_bwd_pre", _p(got), _p(y), _p(dy), mia_hip.BF16, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
                 _p(coefs[4]), n, _c_i64(h * w), c, NORM_INSTANCE, 0, _c_float(0.01), part.shape[1], _p(part), _p(cc[0]), _p(cc[1]),
                 _p(dgb[0]), _p(dgb[1]), _p(dgb[2]), 0, None, _stream())
        else:
            call("mia_norm_act_bwd", _p(got), None, _p(y), _p(dy), mia_hip.BF16, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
                 _p(coefs[4]), n, _c_i64(h * w), c, NORM_INSTANCE, 0, _c_float(0.01), slabs, _p(pt), _p(cc[0]), _p(cc[1]),
                 _p(dgb[0]), _p(dgb[1]), _p(dgb[2]), 0, None, _stream())
        outs.append((dy.float().cpu(), dgb[:2].cpu().clone(), cc.cpu().clone()))
    assert relerr(outs[1][2], outs[0][2]) < 1e-5       # c1, c2
    assert relerr(outs[1][1], outs[0][1]) < 1e-5       # dgamma, dbeta
    assert relerr(outs[1][0], outs[0][0]) < 8e-3       # dy (bf16: a handful of values may round the other way)


@_needs_experiments
@pytest.mark.gpu
def test_fused_backward_reduction_matches_unfused_model():
    """Model level: ops.FUSE_CR on / off on a bf16 UNet with 64-channel level-0 blocks -- logits identical, every parameter
    gradient within the bf16 rounding band of the unfused path (the reduction partials are summed in a different order, which
    can move single dy values by one bf16 ulp)."""
    from losses.compound_losses import DiceAndCELoss
    from mia_hip import ops
    from models.unet import UNet
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    x = torch.rand(3, 1, 80, 112, generator=g).to(dev)
    lab = torch.randint(0, 3, (3, 80, 112), generator=g).to(dev)
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    res = {}
    old = ops.FUSE_CR
    try:
        for fuse in (False, True):
            ops.FUSE_CR = fuse
            ops._CR_HINT.clear()
            torch.manual_seed(11)
            m = UNet(2, 1, 3, [64, 128], normalization="instance", dropout_prob=None).to(dev)
            m.set_compute_dtype(torch.bfloat16)
            m.train()
            out = m(x)
            loss_fn(out, lab).backward()
            res[fuse] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
            assert not ops._CR_HINT  # every hint was consumed
    finally:
        ops.FUSE_CR = old
    assert torch.equal(res[False][0], res[True][0])
    for k, gk in res[True][1].items():
        assert relerr(gk, res[False][1][k]) < 2e-2, k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("norm,ch,hw", [("instance", [32, 64, 128], (64, 96)), ("batch", [64, 128], (48, 80)), ("instance", [16, 32, 64, 128], (80, 48))])
def test_skip_gradient_accumulated_in_the_stride2_input_gradient(norm, ch, hw, dtype):
    """A skip tensor's gradient arrives in two pieces (decoder block, next encoder level).  ops.FUSE_ACC: the stride-2 block adds
    its piece into the decoder's tensor inside its input-gradient kernel (`mia_conv_mma_acc`) and the skip block's norm backward
    reads ONE tensor.  fp32: the stored sum is the same fp32 number the two-piece kernels formed on load -> every gradient
    bit-identical; bf16: the sum is rounded to bf16 once more -> within the bf16 rounding band.  Kernel level: out += conv."""
    import mia_hip
    from mia_hip import CONV_T3S2, call, ops
    from mia_hip.ops import _p, _stream
    from models.unet import UNet
    dev = _dev()
    g = torch.Generator().manual_seed(ch[0] + hw[0])
    x = torch.rand(2, 1, *hw, generator=g).to(dev)
    gz = torch.randn(2, 3, *hw, generator=g).to(dev)
    res = {}
    old, raw_call, calls = ops.FUSE_ACC, ops.call, []

    def counting_call(name, *a):
        calls.append(name)
        return raw_call(name, *a)

    ops.call = counting_call
    try:
        for fuse in (False, True):
            ops.FUSE_ACC = fuse
            del calls[:]
            torch.manual_seed(4)
            m = UNet(2, 1, 3, ch, normalization=norm, dropout_prob=None).to(dev)
            m.set_compute_dtype(dtype)
            m.train()
            (m(x) * gz).sum().backward()
            res[fuse] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
            assert not ops._ACC_HINT or not fuse  # every piece left by a decoder block was taken
            assert calls.count("mia_conv_mma_acc") == (len(ch) - 1 if fuse else 0)  # one accumulating launch per skip level
    finally:
        ops.FUSE_ACC, ops.call = old, raw_call
    for k in res[False]:
        if dtype == torch.float32:
            assert torch.equal(res[False][k], res[True][k]), k
        else:
            assert relerr(res[True][k], res[False][k]) < 3e-2, k
    # kernel level
    n, cout, cin, hc, wc = 2, 64, 32, 12, 20
    dt = mia_hip.BF16 if dtype == torch.bfloat16 else mia_hip.F32
    dy = torch.randn(n, hc, wc, cout, generator=g).to(dev, dtype)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / 30).to(dev)
    wb, npad, kpad = ops.PackCache().get(wt, dt, False)
    for fine in ((2 * hc, 2 * wc), (2 * hc - 1, 2 * wc - 1)):
        base = torch.randn(n, fine[0], fine[1], cin, generator=g).to(dev, dtype)
        plain, _, _ = ops.conv_mma(CONV_T3S2, dy, None, wb, npad, kpad, False, None, cin, fine)
        acc = base.clone()
        call("mia_conv_mma_acc", CONV_T3S2, dt, _p(dy), cout, _p(wb), npad, kpad, 0, _p(acc), cin, n, hc, wc, fine[0], fine[1], None, None, None, _stream())
        want = (base.float() + plain.float()).to(dtype)
        assert torch.equal(acc, want), fine


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,k1", [(torch.float32, 3), (torch.bfloat16, 3), (torch.bfloat16, 4), (torch.float32, 2)])
def test_head_weight_gradient_in_the_norm_backward_reduction_pass(dtype, k1):
    """ops.FUSE_HEAD_W: `mia_norm_act_bwd_head_w` accumulates the 1x1 head's dW / db in the kernel that adds up the last block's
    norm-backward sums (both read exactly dlogits and y) -- `mia_head_norm_wgrad`'s pass over them disappears.  Every other
    gradient bit-identical to the two-pass path; the head's dW / db equal within fp32 summation order (another slab partition)."""
    from losses.compound_losses import DiceAndCELoss
    from mia_hip import ops
    from models.unet import UNet
    dev = _dev()
    g = torch.Generator().manual_seed(k1)
    x = torch.rand(3, 1, 96, 80, generator=g).to(dev)
    lab = torch.randint(0, k1, (3, 96, 80), generator=g).to(dev)
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=k1 - 1, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    res = {}
    old = ops.FUSE_HEAD_W
    try:
        for fuse in (False, True):
            ops.FUSE_HEAD_W = fuse
            torch.manual_seed(2)
            m = UNet(2, 1, k1, [64, 128], normalization="instance", dropout_prob=None).to(dev)
            m.set_compute_dtype(dtype)
            m.train()
            loss_fn(m(x), lab).backward()
            res[fuse] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    finally:
        ops.FUSE_HEAD_W = old
    for k in res[False]:
        if k.startswith("decoder.seg_output"):
            assert relerr(res[True][k], res[False][k]) < 2e-5, k
        else:
            assert torch.equal(res[False][k], res[True][k]), k


# fp32 tensors, products on the f16 matrix cores from two-part split operands (option f32_split, DEFAULT; csrc/common.h SplitF16):
# every operand element, scaled by its tensor's power of two, enters as h + l in fp16 (22-23 significand bits), products are exact,
# accumulation is fp32 and the rescaling is exact -- so a result must be as close to fp64 math as the exact fp32 MFMA kernel's.
def _split_close(got, exact, want, what):
    e_got, e_exact = relerr(got, want), relerr(exact, want)
    assert e_exact < TOL[torch.float32], (what, e_exact)
    assert e_got < 2e-6 and e_got < 2.0 * e_exact + 5e-7, (what, e_got, e_exact)
    assert not torch.equal(got, exact), (what, "the split kernel did not run")


@pytest.fixture
def split_everywhere():
    """ops.F32_SPLIT_MIN_MACS = 0: the size gate (small launches stay on the exact kernel) is lifted so that test-sized shapes run
    the split kernels."""
    from mia_hip import ops
    old = ops.F32_SPLIT_MIN_MACS
    ops.F32_SPLIT_MIN_MACS = 0
    yield
    ops.F32_SPLIT_MIN_MACS = old


@pytest.mark.gpu
@pytest.mark.parametrize("scales", [(1.0, 1.0, 1.0), (3e-7, 40.0, 2e-9), (5e4, 1e-3, 7e5)])
@pytest.mark.parametrize("case", [(2, 64, 0, 64, 40, 56), (1, 128, 0, 256, 24, 40), (1, 64, 64, 64, 33, 47), (2, 32, 0, 32, 36, 52),
                                  (1, 256, 0, 128, 17, 33), (2, 16, 0, 48, 20, 28), (1, 96, 96, 96, 24, 40),
                                  # narrow layers (<= 32 channels everywhere: the 8-row-tile weight gradient), two sources, ragged heights
                                  (2, 32, 32, 32, 20, 44), (1, 16, 16, 32, 19, 21), (3, 32, 0, 16, 9, 70)])
def test_f32_split_conv_and_weight_gradient_have_fp32_accuracy(case, scales, split_everywhere):
    """Option f32_split (default): 3x3 forward (stride 1 and 2), both input gradients and both weight gradients in fp32 on split-f16
    products -- as close to fp64 math as the exact fp32 kernels (max-norm, relative: 2e-6 and within 2x the exact kernel's own
    distance), not bit-equal to them (the split kernels are the ones that ran), statistics included.  `scales` moves activations,
    weights and output gradients far away from 1 (gradient-sized 1e-9, 1e5-sized activations): the per-tensor power-of-two scaling
    must keep fp16's range out of the picture; the two sources of a concatenated input get DIFFERENT magnitudes."""
    import mia_hip
    from mia_hip import ops, CONV_G3S1, CONV_G3S2, CONV_T3S2, WGRAD_3S1, WGRAD_3S2
    dev = _dev()
    n, c1, c2, cout, h, w = case
    sx, sw, sd = scales
    cin = c1 + c2
    g = torch.Generator().manual_seed(cin + cout + h)
    old = mia_hip.get_option("f32_split")
    try:
        for stride in (1, 2):
            x = torch.randn(n, cin, h, w, generator=g) * sx
            if c2:
                x[:, c1:] *= 37.0
            wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9) * sw
            b = torch.randn(cout, generator=g) * (sx * sw)
            ho, wo = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
            dy = torch.randn(n, cout, ho, wo, generator=g) * sd
            xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
            yr = F.conv2d(xr.double(), wr.double(), b.double(), stride=stride, padding=1)
            yr.backward(dy.double())
            x1 = nhwc(x[:, :c1], torch.float32, dev)
            x2 = nhwc(x[:, c1:], torch.float32, dev) if c2 else None
            dyd = nhwc(dy, torch.float32, dev)
            wd = wt.to(dev)
            res = {}
            for flag in (0, 1, 2):
                mia_hip.set_option("f32_split", flag)
                pc = ops.PackCache()
                wp, npad, kpad = pc.get(wd, mia_hip.F32, True)
                wb, npb, kpb = pc.get(wd, mia_hip.F32, False)
                y, _, stats = ops.conv_mma(CONV_G3S2 if stride == 2 else CONV_G3S1, x1, x2, wp, npad, kpad, False, b.to(dev), cout, (ho, wo),
                                           want_stats=True)
                split = c1 if c2 else None
                if stride == 2:
                    dx1, dx2, _ = ops.conv_mma(CONV_T3S2, dyd, None, wb, npb, kpb, False, None, cin, (h, w), out_split=split)
                else:
                    dx1, dx2, _ = ops.conv_mma(CONV_G3S1, dyd, None, wb, npb, kpb, True, None, cin, (h, w), out_split=split)
                dx = nchw(dx1) if dx2 is None else torch.cat([nchw(dx1), nchw(dx2)], 1)
                dw = ops.conv_wgrad(WGRAD_3S2 if stride == 2 else WGRAD_3S1, x1, x2, dyd, wt.shape, cout, cin).cpu()
                res[flag] = (nchw(y), stats.sum(1).cpu(), dx, dw)
            for flag in (1, 2):  # four products on interleaved words; three on planes (32 x 32 tiles, where the shape has them)
                (y0, s0, dx0, dw0), (y1, s1, dx1_, dw1) = res[0], res[flag]
                for name, exact, got, want in (("y", y0, y1, yr), ("dx", dx0, dx1_, xr.grad), ("dw", dw0, dw1, wr.grad)):
                    _split_close(got, exact, want, (name, stride, flag))
                assert relerr(s1[..., 0], yr.detach().sum((2, 3))) < 1e-3
                assert relerr(s1[..., 1], (yr.detach() ** 2).sum((2, 3))) < 1e-3
    finally:
        mia_hip.set_option("f32_split", old)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 128, 64, 20, 28), (1, 64, 32, 33, 17), (1, 512, 256, 8, 12)])
def test_f32_split_transposed_conv_has_fp32_accuracy(case, split_everywhere):
    """Option f32_split on the ConvTranspose 2x2 / stride 2 trio (forward, input gradient, weight gradient) in fp32."""
    import mia_hip
    from mia_hip import ops, CONV_G2S2, CONV_T2S2, WGRAD_2S2
    dev = _dev()
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 2, 2, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    dy = torch.randn(n, cout, 2 * h, 2 * w, generator=g) * 1e-6
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr.double(), wr.double(), b.double(), stride=2)
    yr.backward(dy.double())
    xd, dyd, wd = nhwc(x, torch.float32, dev), nhwc(dy, torch.float32, dev), wt.to(dev)
    old = mia_hip.get_option("f32_split")
    res = {}
    try:
        for flag in (0, 1, 2):
            mia_hip.set_option("f32_split", flag)
            pc = ops.PackCache()
            wf, nf, kf = pc.get(wd, mia_hip.F32, False)
            wb, nb, kb = pc.get(wd, mia_hip.F32, True)
            y, _, _ = ops.conv_mma(CONV_T2S2, xd, None, wf, nf, kf, False, b.to(dev), cout, (2 * h, 2 * w))
            dx, _, _ = ops.conv_mma(CONV_G2S2, dyd, None, wb, nb, kb, False, None, cin, (h, w))
            dw = ops.conv_wgrad(WGRAD_2S2, dyd, None, xd, wt.shape, cin, cout).cpu()
            res[flag] = (nchw(y), nchw(dx), dw)
    finally:
        mia_hip.set_option("f32_split", old)
    for flag in (1, 2):
        for name, exact, got, want in zip(("y", "dx", "dw"), res[0], res[flag], (yr, xr.grad, wr.grad)):
            _split_close(got, exact, want, (name, flag))


@pytest.mark.gpu
@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("c1,c2,cout", [(32, 0, 96), (64, 32, 64), (16, 0, 24)])
def test_amax_by_products_equal_the_tensors_maxima(c1, c2, cout, norm, split_everywhere):
    """The maxima the split-f16 convs scale by are by-products of the kernels that WRITE the operands: the forward apply pass (z), the
    backward apply pass (dy), the ConvTranspose epilogue (its output) and the two-destination input gradient (dx2).  Each slot must
    hold exactly the bit pattern of max |tensor| (96 and 24 channels leave idle lanes / scalar fallbacks in the streaming kernels),
    and a full block forward + backward must launch NO standalone `mia_amax` for them."""
    import mia_hip
    from mia_hip import ops, NORM_BATCH, NORM_INSTANCE, CONV_T2S2
    dev = _dev()
    g = torch.Generator().manual_seed(c1 + cout)
    n, h, w = 2, 20, 36
    x1 = (torch.randn(n, h, w, c1, generator=g) * 3.0).to(dev).requires_grad_(True)
    x2 = (torch.randn(n, h, w, c2, generator=g) * 0.02).to(dev).requires_grad_(True) if c2 else None
    wt = torch.nn.Parameter((torch.randn(cout, c1 + c2, 3, 3, generator=g) / math.sqrt(9 * (c1 + c2))).to(dev))
    b, gm, bt = (torch.nn.Parameter(torch.randn(cout, generator=g).to(dev)) for _ in range(3))
    cfg = ops.NormCfg(NORM_BATCH if norm == "batch" else NORM_INSTANCE, True, running_mean=torch.zeros(cout, device=dev),
                      running_var=torch.ones(cout, device=dev), num_batches=torch.zeros((), dtype=torch.long, device=dev))
    seen = {}
    raw = ops.call

    def spy(name, *a):
        if name == "mia_amax":
            seen["amax"] = seen.get("amax", 0) + 1
        return raw(name, *a)

    def bits(t):
        return t.detach().abs().max().view(torch.int32).item()

    ops.amax_slot(x1), ops.amax_slot(wt)  # (inputs of the test itself: measured up front, outside the count)
    if x2 is not None:
        ops.amax_slot(x2)
    ops.pack_cache(wt).get(wt, mia_hip.F32, True), ops.pack_cache(wt).get(wt, mia_hip.F32, False)
    ops.call = spy
    try:
        z = ops.PlainBlockFn.apply(x1, x2, wt, b, gm, bt, 1, cfg)
        assert z._mia_amax[0].item() == bits(z) and z._mia_amax[1] == z._version
        dz = torch.randn(z.shape, generator=g).to(dev) * 1e-5
        captured = {}
        orig_wgrad = ops.conv_wgrad

        def wgrad_spy(mode, xa, xb, dy, *a, **k):
            captured["dy"] = dy
            return orig_wgrad(mode, xa, xb, dy, *a, **k)

        ops.conv_wgrad = wgrad_spy
        try:
            z.backward(dz)
        finally:
            ops.conv_wgrad = orig_wgrad
        dy = captured["dy"]
        assert dy._mia_amax[0].item() == bits(dy)
        assert seen.get("amax", 0) == 0, seen
    finally:
        ops.call = raw
    # ConvTranspose epilogue and the second destination of an input gradient
    xt = torch.randn(n, 9, 13, 64, generator=g).to(dev)
    wT = (torch.randn(64, 32, 2, 2, generator=g) / 8.0).to(dev)
    wf, nf, kf = ops.PackCache().get(wT, mia_hip.F32, False)
    up, _, _ = ops.conv_mma(CONV_T2S2, xt, None, wf, nf, kf, False, torch.randn(32, generator=g).to(dev), 32, (18, 26))
    assert up._mia_amax[0].item() == bits(up)
    if c2:
        from mia_hip import CONV_G3S1
        wb, npb, kpb = ops.pack_cache(wt).get(wt, mia_hip.F32, False)
        d1, d2, _ = ops.conv_mma(CONV_G3S1, dy, None, wb, npb, kpb, True, None, c1 + c2, (h, w), out_split=c1)
        assert d2._mia_amax[0].item() == bits(d2)


@pytest.mark.gpu
def test_f32_split_wide_dynamic_range_inside_one_tensor(split_everywhere):
    """One operand tensor whose images differ by 2^-16 in magnitude (one sample's activations / gradients dwarf another's): the
    per-tensor scale is set by the largest, the small image's elements keep >= 17 significand bits (fp16 denormal low parts; the f16
    MFMA keeps them -- tools/probe/mfma_f16_denorm.hip), so ITS outputs stay within 1e-5 of fp64 math relative to their own size and
    the large image's within the fp32 bar.  `mia_amax` itself is checked against torch (bit pattern of max |x|, NaN visible)."""
    import mia_hip
    from mia_hip import ops, CONV_G3S1, call
    from mia_hip.ops import _c_i64, _p, _stream
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 24, 40, generator=g)
    x[1] *= 2.0 ** -16
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    want = F.conv2d(x.double(), wt.double(), padding=1)
    xd, wd = nhwc(x, torch.float32, dev), wt.to(dev)
    wp, npad, kpad = ops.PackCache().get(wd, mia_hip.F32, True)
    y, _, _ = ops.conv_mma(CONV_G3S1, xd, None, wp, npad, kpad, False, None, 64, (24, 40))
    y = nchw(y)
    assert relerr(y[0], want[0]) < 2e-6 and relerr(y[1], want[1]) < 1e-5, (relerr(y[0], want[0]), relerr(y[1], want[1]))
    slot = ops.amax_slot(xd)
    assert slot.item() == x.abs().max().view(torch.int32).item()
    for n_el in (1, 3, 5, 1023, 4096 + 7):
        t = torch.randn(n_el + 1, generator=g).to(dev)[1:]  # a 4-byte-aligned, not 16-byte-aligned start
        s2 = torch.full((1,), 123, device=dev, dtype=torch.int32)
        call("mia_amax", _p(t), _c_i64(n_el), _p(s2), 1, _stream())
        assert s2.item() == t.abs().max().view(torch.int32).item(), n_el
    t = torch.randn(1000, generator=g)
    t[77] = float("nan")
    assert math.isnan(ops.amax_slot(t.to(dev)).view(torch.float32).item())


@pytest.mark.gpu
def test_batched_weight_pack_matches_per_tensor_pack():
    """`mia_pack_weight_batch` (ops.PackPlan: one launch re-packs every weight of a model after the optimizer step) against the
    per-tensor `mia_pack_weight` on the same values: both orientations, 3x3 / 2x2 / 1x1 taps, channel counts that are not multiples
    of the 16 x 64 brick (96, 40, 3), bf16 and fp32 -- bit-identical, zero padding included."""
    import mia_hip
    from mia_hip import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 64, 3, 3), (96, 192, 3, 3), (128, 40, 3, 3), (256, 128, 2, 2), (96, 48, 2, 2), (3, 64, 1, 1), (16, 16, 3, 3), (1024, 512, 3, 3)]
    for dt in (mia_hip.BF16, mia_hip.F32):
        ws = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
        for w in ws:  # first use: per-tensor packs, which also tells the plan which copies exist
            for orient in (True, False):
                ops.pack_cache(w).get(w, dt, n_from_d0=orient)
        plan = ops.PackPlan(ws, dt)
        with torch.no_grad():
            for w in ws:
                w.mul_(1.7).add_(0.01)  # new values behind the caches' back
        ops.bump_param_epoch()
        plan.repack()
        torch.cuda.synchronize()
        for w in ws:
            for orient in (True, False):
                got, npad, kpad = ops.pack_cache(w).get(w, dt, n_from_d0=orient)   # planted by the plan
                ref, npad2, kpad2 = ops.PackCache().get(w, dt, n_from_d0=orient)   # a fresh per-tensor pack of the same values
                assert (npad, kpad) == (npad2, kpad2)
                assert got.data_ptr() != ref.data_ptr() and torch.equal(got, ref), (tuple(w.shape), orient, dt)


