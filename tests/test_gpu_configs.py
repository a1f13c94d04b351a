"""BASELINE.json configs at their own widths against the CPU oracle.

* cfg2/cfg3 model `[64..1024]`: one fp32 TRAIN STEP (forward, Dice+CE, every parameter gradient, clip, AdamW) vs
  `oracle/train_ref.train_step` -- the first place the branch-free conv / dgrad / wgrad kernels meet the oracle end to
  end -- and the same model in bf16 (the benchmark's dtype) vs the fp32 oracle with stated tolerances;
* cfg4: `al_train` default widths `[32..512]`, 256x256, K1 = 3, batch norm, fed by `BatchedAugment(al_train_transforms("busi"))`
  vs the oracle's pipeline + train step;
* cfg5: `[96..3072]` six levels: one fp32 image vs the oracle at a size the CPU affords, and the full 768x768 bs 16 bf16
  shape through size-independent properties (batch independence, determinism, descent).

Tolerances: fp32 logits / loss 1e-4 (north_star); fp32 parameter gradients 2e-3 of each tensor's max (accumulation order
differs: tiles + split-K slabs vs oneDNN); bf16 -- see `BF16_*` below.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# bf16 activations (8 significant bits, relative rounding 2^-9 per stored tensor) through 23 conv layers with fp32
# accumulation, statistics, parameters and loss.  Measured on MI355X for the [64..1024] model at 128x128 (this test prints
# the figures): logits max |err| ~1.2e-2 of the logit range, loss ~2e-3, per-tensor gradient relative L2 error 1-4 %
# (largest on the first encoder convs, which sit behind the whole backward chain).  The bounds leave 2-3x headroom over
# those figures and stay far below what a structural fault produces: a dropped 16x16 output tile of a 128x128 map is
# >= 12 % relative L2 on that layer's weight gradient, a wrong / missing tap ~33 %, a missed channel chunk >= 50 %.
BF16_LOGIT_TOL = 4e-2   # max |logit - oracle| / (max - min of oracle logits)
BF16_LOSS_TOL = 1e-2    # |loss - oracle loss|
BF16_GRAD_REL_L2 = 0.10  # ||g - g_oracle||_2 / ||g_oracle||_2 per parameter tensor (weights of convs)
BF16_GRAD_COS = 0.995   # and the direction


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _batch(n, s, seed=1337, k1=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 1, s, s, generator=g)
    yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), indexing="ij")
    lab = torch.zeros(n, s, s, dtype=torch.long)
    for i in range(n):
        cy, cx, ry, rx = (torch.rand(4, generator=g) * torch.tensor([s / 2, s / 2, s / 6, s / 6]) +
                          torch.tensor([s / 4, s / 4, s / 12, s / 12])).tolist()
        lab[i][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = 1
        if k1 > 2:
            lab[i][((yy - cy - ry) / (ry / 2)) ** 2 + ((xx - cx) / (rx / 2)) ** 2 < 1] = 2
    return x, lab


def _model(dev, channels, norm, k1=3, dtype=torch.float32, seed=1337):
    from models.unet import UNet
    torch.manual_seed(seed)
    m = UNet(2, 1, k1, channels, normalization=norm, dropout_prob=None).to(dev)
    if dtype == torch.bfloat16:
        m.set_compute_dtype(torch.bfloat16)
    return m


def _loss_fn(k1):
    from losses.compound_losses import DiceAndCELoss
    from losses.dice_loss import DiceLoss
    return DiceAndCELoss(dice_loss=DiceLoss, dice_kwargs=dict(num_classes=k1 - 1, smooth=1e-5, do_bg=True, softmax=True,
                                                              batch=False, squared=False),
                         ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs={})


def _oracle_step(state, x, y, k1, norm, lr, opt_name="adamw", wd=5e-4):
    """oracle/train_ref.train_step on a copy of `state`; returns logits, loss, grads (pre-clip), grad norm, post-step state."""
    from oracle import train_ref
    params = {k: v.detach().clone() for k, v in state.items()}
    opt = train_ref.make_optimizer(params, opt_name, weight_decay=wd)
    # gradients before clipping: clip_grad_norm_ scales .grad in place, so run the pieces of train_step by hand first
    from oracle import losses_ref, unet_ref
    out = unet_ref.unet_forward(params, x.float(), norm, True)
    loss = losses_ref.dice_and_ce(out, y.long(), k1 - 1)
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in train_ref.trainable(params).items()}
    for v in params.values():
        v.grad = None
    # reset batch-norm running statistics touched by the first forward, then the real step
    params2 = {k: v.detach().clone() for k, v in state.items()}
    opt2 = train_ref.make_optimizer(params2, opt_name, weight_decay=wd)
    res = train_ref.train_step(params2, opt2, x, y, k1 - 1, normalization=norm, lr=lr, max_grad_norm=10.0)
    return out.detach(), float(loss), grads, float(res["grad_norm"]), {k: v.detach() for k, v in params2.items()}


def _fp32_step_vs_oracle(channels, norm, size, n, k1=3, seed=3, lr=1e-3):
    from oracle import train_ref
    dev = _dev()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    m = _model(dev, channels, norm, k1).train()
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, y = _batch(n, size, seed=seed, k1=k1)
    ref_logits, ref_loss, ref_grads, ref_gn, ref_post = _oracle_step(state, x, y, k1, norm, lr)
    loss_fn = _loss_fn(k1)
    opt = torch.optim.AdamW(m.parameters(), betas=(0.9, 0.999), weight_decay=5e-4)
    for g in opt.param_groups:
        g["lr"] = lr
    out = m(x.to(dev))
    loss = loss_fn(out, y.to(dev))
    assert float((out.detach().cpu() - ref_logits).abs().max()) < 1e-4
    assert abs(loss.item() - ref_loss) < 1e-4
    top2 = ref_logits.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2e-4
    assert (out.detach().cpu().argmax(1)[safe] == ref_logits.argmax(1)[safe]).all()  # label maps, bit-exact off ties
    opt.zero_grad()
    loss.backward()
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        ref = ref_grads[name]
        err = float((p.grad.cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-3))
        worst = max(worst, (name, err), key=lambda t: t[1])
        assert err < 2e-3, (name, err)
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=10.0)
    assert abs(gn.item() - ref_gn) / ref_gn < 1e-3
    opt.step()
    for k, v in m.state_dict().items():
        ref = ref_post[k]
        if ref.is_floating_point():
            tol = 1e-5 if "running" in k else 2.5 * lr + 1e-6
            assert float((v.cpu() - ref).abs().max()) < tol, k
        else:
            assert torch.equal(v.cpu(), ref), k
    print(f"[fp32 {channels[0]}..{channels[-1]} {norm} {size}x{size}x{n}] loss {loss.item():.6f} (oracle {ref_loss:.6f}), "
          f"worst grad err {worst[1]:.2e} at {worst[0]}")


def test_full_width_fp32_train_step_vs_oracle():
    """[64..1024] (cfg2 / cfg3 model), 2 images of 128x128: every conv on the path takes the branch-free kernels
    (c % 32 == 0), forward AND backward, against oracle/train_ref (al_trainer.py:1350-1399)."""
    _fp32_step_vs_oracle([64, 128, 256, 512, 1024], "instance", 128, 2)


def test_full_width_bf16_train_step_vs_fp32_oracle():
    """The benchmark's dtype at the benchmark's widths vs the fp32 CPU oracle: logits, loss, every weight gradient."""
    from oracle import train_ref
    dev = _dev()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    channels, k1, norm = [64, 128, 256, 512, 1024], 3, "instance"
    m = _model(dev, channels, norm, k1, torch.bfloat16).train()
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, y = _batch(2, 128, seed=3)
    ref_logits, ref_loss, ref_grads, _, _ = _oracle_step(state, x, y, k1, norm, 1e-3)
    out = m(x.to(dev))
    loss = _loss_fn(k1)(out, y.to(dev))
    rng = float(ref_logits.max() - ref_logits.min())
    lerr = float((out.detach().cpu() - ref_logits).abs().max()) / rng
    assert lerr < BF16_LOGIT_TOL, lerr
    assert abs(loss.item() - ref_loss) < BF16_LOSS_TOL, (loss.item(), ref_loss)
    agree = float((out.detach().cpu().argmax(1) == ref_logits.argmax(1)).float().mean())
    loss.backward()
    worst_l2, worst_cos = ("", 0.0), ("", 1.0)
    for name, p in m.named_parameters():
        ref = ref_grads[name].double()
        got = p.grad.cpu().double()
        if float(ref.norm()) < 1e-6:
            continue
        rel = float((got - ref).norm() / ref.norm())
        cos = float((got * ref).sum() / (got.norm() * ref.norm() + 1e-30))
        worst_l2 = max(worst_l2, (name, rel), key=lambda t: t[1])
        worst_cos = min(worst_cos, (name, cos), key=lambda t: t[1])
        if name.endswith("all.0.weight") or "upsamples" in name and name.endswith("weight") or "seg_output" in name:
            assert rel < BF16_GRAD_REL_L2, (name, rel)
            assert cos > BF16_GRAD_COS, (name, cos)
        else:  # norm affine / bias vectors: few elements, sums over whole maps -- same bound on the direction only
            assert cos > 0.98, (name, cos)
    print(f"[bf16 64..1024 128x128x2] logit err {lerr:.2e} of range, loss {loss.item():.5f} vs {ref_loss:.5f}, argmax agree "
          f"{agree:.5f}, worst rel-L2 {worst_l2[1]:.3f} at {worst_l2[0]}, worst cos {worst_cos[1]:.5f} at {worst_cos[0]}")


def test_cfg4_busi_pipeline_and_train_step_vs_oracle():
    """cfg4: al_train default model `[32,64,128,256,512]`, K1 = 3, batch norm (train.py:25), 256x256, fed by the on-GPU
    augmentation pipeline (al_trainer.py:670-697 -> JointResize 256) -- vs the oracle pipeline + train step, bs 2."""
    from oracle import transforms_ref as R
    from oracle import train_ref
    from transforms.gpu_pipeline import BatchedAugment, al_train_transforms
    from transforms.image_transform import RandomGaussianNoise
    dev = _dev()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    h0, w0, s, k1, norm, lr = 300, 364, 256, 3, "batch", 1e-3
    g = torch.Generator().manual_seed(21)
    imgs = torch.rand(2, 1, h0, w0, generator=g)
    _, labs = _batch(2, max(h0, w0), seed=22)
    labs = labs[:, :h0, :w0].contiguous()
    RandomGaussianNoise.exact_rng = True  # noise tensor from the host generator like the reference (image_transform.py:130)
    try:
        seed = None
        for cand in range(200):  # a seed whose draws exercise at least three stages incl. a geometric one, on both samples
            torch.manual_seed(5000 + cand)
            recs = [R.al_train_fugc_pipeline(imgs[i].clone(), labs[i:i + 1].clone())[2] for i in range(2)]
            names = [n for r in recs for n, _ in r]
            if len(names) >= 3 and "affine" in names:
                seed = 5000 + cand
                break
        assert seed is not None
        torch.manual_seed(seed)
        ref_items = [R.al_train_fugc_pipeline(imgs[i].clone(), labs[i:i + 1].clone()) for i in range(2)]
        ref_x = torch.stack([R.apply_resize_image(ri, (s, s), antialias=True) for ri, _, _ in ref_items])
        ref_y = torch.stack([R.apply_resize_label(rl, (s, s))[0] for _, rl, _ in ref_items])
        torch.manual_seed(seed)
        batch = BatchedAugment(al_train_transforms("busi"), image_size=s, do_normalize=False)(imgs.to(dev), labs.to(dev))
    finally:
        RandomGaussianNoise.exact_rng = False
    assert batch["image"].shape == (2, 1, s, s) and batch["label"].shape == (2, s, s)
    assert float((batch["image"].cpu() - ref_x).abs().max()) < 2e-5
    assert torch.equal(batch["label"].cpu(), ref_y)
    # train step on the augmented batch (HIP batch -> HIP model, oracle batch -> oracle model)
    channels = [32, 64, 128, 256, 512]
    m = _model(dev, channels, norm, k1).train()
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref_logits, ref_loss, ref_grads, ref_gn, ref_post = _oracle_step(state, ref_x, ref_y, k1, norm, lr)
    from training.engine import TrainEngine
    eng = TrainEngine(m, _loss_fn(k1), "adamw", {"weight_decay": 5e-4}, start_lr=lr, num_iters=4000, lr_warmup_iter=0)
    out = m(batch["image"])
    assert float((out.detach().cpu() - ref_logits).abs().max()) < 1e-4
    loss = eng.train_step(batch)
    assert abs(loss.item() - ref_loss) < 1e-4
    assert abs(eng.optimizer.last_norm[0].item() - ref_gn) / ref_gn < 1e-3
    for name, p in m.named_parameters():
        ref = ref_grads[name]
        err = float((p.grad.cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-3))
        assert err < 2e-3, (name, err)
    for k, v in m.state_dict().items():
        ref = ref_post[k]
        if "running" in k:
            assert float((v.cpu() - ref).abs().max()) < 2e-5, k
        elif ref.is_floating_point():
            assert float((v.cpu() - ref).abs().max()) < 2.5 * lr + 1e-6, k
    print(f"[cfg4] stages {names}, loss {loss.item():.6f} (oracle {ref_loss:.6f})")


CH5 = [96, 192, 384, 768, 1536, 3072]


def test_cfg5_fp32_train_step_vs_oracle_small():
    """cfg5 widths `[96..3072]`, six levels, one 96x96 image in fp32 vs the oracle (3x3 bottleneck; 279.8 M parameters)."""
    _fp32_step_vs_oracle(CH5, "instance", 96, 1, seed=9)


def test_cfg5_full_size_bf16_properties():
    """cfg5 at its own shape -- 768x768, batch 16, bf16 -- through size-independent properties: instance-norm logits of 16
    images == the same images in groups of 4 (bit-exact), a train step's gradients are run-to-run identical, a few steps
    lower the loss, everything finite."""
    from training.engine import TrainEngine
    dev = _dev()
    x, y = _batch(16, 768, seed=5)
    batch = {"image": x.to(dev), "label": y.to(dev)}
    m = _model(dev, CH5, "instance", 3, torch.bfloat16)
    m.eval()
    with torch.no_grad():
        full = m(batch["image"]).clone()
        for i in range(0, 16, 4):
            assert torch.equal(m(batch["image"][i:i + 4]), full[i:i + 4]), i
    assert full.shape == (16, 3, 768, 768) and torch.isfinite(full).all()
    del full
    loss_fn = _loss_fn(3)
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
    del grads
    losses = [eng.train_step(batch).item() for _ in range(5)]
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
