#!/usr/bin/env python
"""Convergence sanity run: a few hundred TrainEngine steps on a small fixed synthetic set (ellipse masks),
bf16 and fp32, reporting loss and hard Dice of the predicted label maps.  Not a benchmark; a does-it-train check.

    python tools/train_sanity.py [--steps 300 --size 128 --images 64 --dtype bf16|f32|both]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def make_set(n, s, seed=1337):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), indexing="ij")
    img = torch.zeros(n, 1, s, s)
    lab = torch.zeros(n, s, s, dtype=torch.long)
    for i in range(n):
        cy, cx, ry, rx = (torch.rand(4, generator=g) * torch.tensor([s / 2, s / 2, s / 6, s / 6]) +
                          torch.tensor([s / 4, s / 4, s / 10, s / 10])).tolist()
        m1 = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        m2 = ((yy - cy - 0.6 * ry) / (0.5 * ry)) ** 2 + ((xx - cx) / (0.5 * rx)) ** 2 < 1
        lab[i][m1] = 1
        lab[i][m2] = 2
        img[i, 0] = 0.25 + 0.35 * m1.float() + 0.3 * m2.float() + 0.08 * torch.randn(s, s, generator=g)
    return img.clamp(0, 1), lab


def run(dtype, steps, size, nimg, batch):
    from losses.compound_losses import DiceAndCELoss
    from metric.segmentation import predict_and_dice
    from models.unet import UNet
    from training.engine import TrainEngine
    dev = torch.device("cuda:0")
    torch.manual_seed(1337)
    model = UNet(2, 1, 3, [32, 64, 128, 256, 512], normalization="batch", dropout_prob=0.1).to(dev)
    model.set_compute_dtype(dtype)
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, smooth=1e-5, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    eng = TrainEngine(model, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=steps, lr_warmup_iter=steps // 16)
    img, lab = make_set(nimg, size)
    img, lab = img.to(dev), lab.to(dev)
    g = torch.Generator().manual_seed(7)
    for it in range(steps):
        idx = torch.randperm(nimg, generator=g)[:batch].to(dev)
        loss = eng.train_step({"image": img[idx], "label": lab[idx]})
        if it % max(1, steps // 6) == 0 or it == steps - 1:
            print(f"  {str(dtype):15s} step {it:4d} loss {loss.item():.4f}", flush=True)
    model.eval()
    with torch.no_grad():
        _, dice, counts = predict_and_dice(model(img), lab)
    fg = dice[:, 1:].mean().item()
    print(f"  {str(dtype):15s} hard Dice (foreground classes, training images, eval mode): {fg:.4f}")
    return loss.item(), fg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="both")
    a = ap.parse_args()
    res = {}
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        if a.dtype in (name, "both"):
            res[name] = run(dt, a.steps, a.size, a.images, a.batch)
    ok = all(v[1] > 0.9 and v[0] == v[0] for v in res.values())
    print("RESULT", res, "OK" if ok else "NOT CONVERGED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
