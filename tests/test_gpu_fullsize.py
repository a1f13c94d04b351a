"""Parity at BASELINE.json's full sizes (UNet [64..1024], 512x512, batch 32 -- cfg3) through size-independent
properties, plus one direct oracle comparison at full width:

* linearity and shift-equivariance of the canonical 64 -> 64 3x3 conv launch (tile addressing at scale, bit-exact);
* batch independence: instance-norm logits of 32 images == the same images pushed through in groups of 4 (bit-exact);
* run-to-run determinism of a full training step's gradients (split-K weight gradients use a fixed reduction order);
* Dice+CE closed forms at 32 x 3 x 512 x 512 (uniform logits -> CE = ln K1; perfect prediction -> Dice = 0);
* a few training steps on a fixed batch lower the loss and keep every gradient finite;
* fp32 logits of the full-width model on one 256x256 image vs the CPU oracle (1e-4, the north_star tolerance).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CH = [64, 128, 256, 512, 1024]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _conv64(x_nhwc, w, b, dtype):
    from mia_hip import CONV_G3S1, ops
    pc = ops.PackCache()
    wp, npad, kpad = pc.get(w, ops._dt(dtype), True)
    n, h, ww, _ = x_nhwc.shape
    y, _, _ = ops.conv_mma(CONV_G3S1, x_nhwc, None, wp, npad, kpad, False, b, 64, (h, ww))
    return y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_linearity_and_shift_equivariance_full_size(dtype):
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    n, s = 32, 512
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    zero_b = torch.zeros_like(b)
    # inputs and weights in {-1, 0, 1}: every partial sum and every output (|.| < 256, an 11-sigma bound) is an integer
    # that bf16 holds exactly, so linearity must be bit-exact in both dtypes
    xa = torch.randint(-1, 2, (n, s, s, 64), generator=g).to(dev).to(dtype)
    xb = torch.randint(-1, 2, (n, s, s, 64), generator=g).to(dev).to(dtype)
    wi = torch.randint(-1, 2, (64, 64, 3, 3), generator=g).float().to(dev)
    ya, yb, yab = _conv64(xa, wi, zero_b, dtype), _conv64(xb, wi, zero_b, dtype), _conv64(xa + xb, wi, zero_b, dtype)
    assert torch.equal(yab.float(), ya.float() + yb.float())
    del ya, yb, yab, xb
    # shift by one tile (16 px) in x and y: the interior must be bit-identical (same per-pixel summation order)
    x = torch.randn(n, s, s, 64, generator=g).to(dev).to(dtype)
    y = _conv64(x, w, b, dtype)
    xs = torch.roll(x, shifts=(16, 16), dims=(1, 2))
    ys = _conv64(xs, w, b, dtype)
    assert torch.equal(ys[:, 17:-1, 17:-1], y[:, 1:-17, 1:-17])
    assert torch.isfinite(y.float()).all()


def _model(dev, dtype, norm="instance"):
    from models.unet import UNet
    torch.manual_seed(1337)
    m = UNet(2, 1, 3, CH, normalization=norm, dropout_prob=None).to(dev)
    if dtype == torch.bfloat16:
        m.set_compute_dtype(torch.bfloat16)
    return m


def _batch(n, s, seed=1337):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 1, s, s, generator=g)
    yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), indexing="ij")
    lab = torch.zeros(n, s, s, dtype=torch.long)
    for i in range(n):
        cy, cx, ry, rx = (torch.rand(4, generator=g) * torch.tensor([s / 2, s / 2, s / 6, s / 6]) + torch.tensor([s / 4, s / 4, s / 12, s / 12])).tolist()
        lab[i][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = 1
        lab[i][((yy - cy - ry) / (ry / 2)) ** 2 + ((xx - cx) / (rx / 2)) ** 2 < 1] = 2
    return x, lab


def test_batch_independence_full_size_bf16():
    dev = _dev()
    m = _model(dev, torch.bfloat16).eval()
    x, _ = _batch(32, 512)
    x = x.to(dev)
    with torch.no_grad():
        full = m(x).clone()
        for i in range(0, 32, 8):
            part = m(x[i:i + 8])
            assert torch.equal(part, full[i:i + 8]), i
    assert full.shape == (32, 3, 512, 512) and torch.isfinite(full).all()


def test_train_step_determinism_and_descent_full_size_bf16():
    from losses.compound_losses import DiceAndCELoss
    from training.engine import TrainEngine
    dev = _dev()
    x, y = _batch(32, 512)
    batch = {"image": x.to(dev), "label": y.to(dev)}
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    grads = []
    for _ in range(2):  # same weights, same batch -> bit-identical flat gradient (no atomics anywhere on the path)
        m = _model(dev, torch.bfloat16)
        eng = TrainEngine(m, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=100, lr_warmup_iter=0)
        eng.model.train()
        out = eng.model(batch["image"])
        loss = loss_fn(out, batch["label"])
        eng.optimizer.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        grads.append(eng.optimizer.flat_grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    losses = [eng.train_step(batch).item() for _ in range(6)]
    assert all(math.isfinite(v) for v in losses)
    assert losses[-1] < losses[0], losses


def test_loss_closed_forms_full_size():
    from losses.compound_losses import DiceAndCELoss
    from losses.dice_loss import DiceLoss
    dev = _dev()
    _, y = _batch(32, 512)
    y = y.to(dev)
    comp = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    uni = torch.zeros(32, 3, 512, 512, device=dev)
    assert abs(comp.get_ce_loss(uni, y).item() - math.log(3)) < 1e-6
    perfect = torch.nn.functional.one_hot(y, 3).permute(0, 3, 1, 2).float() * 100
    assert abs(DiceLoss(2, do_bg=True)(perfect, y).item()) < 1e-5
    assert abs(comp.get_ce_loss(perfect, y).item()) < 1e-6
    # Dice of uniform probabilities in closed form: 1 - mean_{b,k} (2*T/3 + s) / (HW/3 + T + s)
    hw = 512 * 512
    t = torch.stack([(y == k).sum((1, 2)) for k in range(3)], 1).double().cpu()
    want = (1 - (2 * t / 3 + 1e-5) / (hw / 3 + t + 1e-5)).mean().item()
    assert abs(DiceLoss(2, do_bg=True)(uni, y).item() - want) < 1e-5


def test_full_width_fp32_logits_vs_oracle_256():
    """Direct oracle comparison at the benchmark's widths [64..1024] (one 256x256 image, fp32, eval)."""
    from oracle import unet_ref
    dev = _dev()
    m = _model(dev, torch.float32).eval()
    x, _ = _batch(1, 256, seed=7)
    with torch.no_grad():
        got = m(x.to(dev)).cpu()
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        want = unet_ref.unet_forward(params, x, normalization="instance", training=False)
    assert float((got - want).abs().max()) < 1e-4
    top2 = want.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2e-4
    assert (got.argmax(1)[safe] == want.argmax(1)[safe]).all()


def test_cfg2_full_size_fp32_properties():
    """cfg2 at its own shape -- [64..1024], 256x256, batch 32, fp32 -- through size-independent properties:
    instance-norm logits of 32 images == the same images in groups of 8 (bit-exact), one image of the batch against the CPU
    oracle (1e-4), a train step's gradients run-to-run identical, a few steps lower the loss, everything finite."""
    from losses.compound_losses import DiceAndCELoss
    from oracle import unet_ref
    from training.engine import TrainEngine
    dev = _dev()
    x, y = _batch(32, 256, seed=11)
    batch = {"image": x.to(dev), "label": y.to(dev)}
    m = _model(dev, torch.float32).eval()
    with torch.no_grad():
        full = m(batch["image"]).clone()
        for i in range(0, 32, 8):
            assert torch.equal(m(batch["image"][i:i + 8]), full[i:i + 8]), i
    assert full.shape == (32, 3, 256, 256) and torch.isfinite(full).all()
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        want = unet_ref.unet_forward(params, x[17:18], normalization="instance", training=False)
    got = full[17:18].cpu()
    assert float((got - want).abs().max()) < 1e-4
    top2 = want.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2e-4
    assert (got.argmax(1)[safe] == want.argmax(1)[safe]).all()
    del full
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    eng = TrainEngine(m, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=100, lr_warmup_iter=0)
    grads = []
    for _ in range(2):
        m.train()
        loss = loss_fn(m(batch["image"]), batch["label"])
        eng.optimizer.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        grads.append(eng.optimizer.flat_grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    losses = [eng.train_step(batch).item() for _ in range(5)]
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
