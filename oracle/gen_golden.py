"""Generate tests/golden/*.npz from the IMPORTED REFERENCE (dev container only).

    python -m oracle.gen_golden

Every array written here is an input or an output of the reference's own code
(`/root/reference/src/models/unet`, `losses`, `scheduler`, `transforms`) run on
PyTorch-CPU fp32.  The vectors are data; no reference source travels.
TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _refload

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy().copy()


def _labels(gen, b, h, w, k1):
    """Blobby label maps: nearest-upsampled coarse random classes."""
    coarse = torch.randint(0, k1, (b, 1, h // 4, w // 4), generator=gen).float()
    return torch.nn.functional.interpolate(coarse, size=(h, w), mode="nearest")[:, 0].long()


def gen_unet(ref, tag, normalization, channels, size, batch=2, k1=3, block_type="plain",
             deep_supervision=False, ds_layer=0, train_step=True):
    torch.manual_seed(1337)
    kw = dict(normalization=normalization) if block_type == "plain" else dict(norm_key=normalization)
    model = ref.unet.UNet(2, 1, k1, channels, deep_supervision=deep_supervision, ds_layer=ds_layer,
                          block_type=block_type, dropout_prob=None, **kw)
    gen = torch.Generator().manual_seed(7)
    # non-trivial affine params so gamma/beta paths are exercised
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1 and (".all.2." in n or ".all.1." in n or "downsample_skip.1" in n):
                p.add_(0.2 * torch.randn(p.shape, generator=gen))
    x = torch.rand(batch, 1, size, size, generator=gen)
    y = _labels(gen, batch, size, size, k1)
    d = {"x": _np(x), "labels": _np(y)}
    for k, v in model.state_dict().items():
        d["init/" + k] = _np(v)

    # eval-mode outputs (before any running-stat update)
    model.eval()
    with torch.no_grad():
        lo = model(x)
        d["eval/logits"] = _np(lo)
        d["eval/argmax"] = _np(lo.softmax(1).argmax(1))
        d["eval/enc_feature"] = _np(model.get_enc_feature(x))
        seg, feat = model.get_pixel_feature(x)
        d["eval/pixel_feature"] = _np(feat)
        if deep_supervision:
            outs = model(x, return_ds=True)
            for i, o in enumerate(outs):
                d[f"eval/ds{i}"] = _np(o)

    if train_step:
        model.train()
        loss_fn = ref.compound.DiceAndCELoss(
            dice_loss=ref.dice_loss.DiceLoss,
            dice_kwargs=dict(num_classes=k1 - 1, smooth=1e-5, do_bg=True, softmax=True, batch=False, squared=False),
            ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs={})
        opt = torch.optim.AdamW(model.parameters(), betas=(0.9, 0.999), weight_decay=5e-4)
        sched = ref.lr_scheduler.PolyLRScheduler(opt, initial_lr=1e-3, max_steps=4000, warmup_steps=250)
        sched.step(0)
        d["train/lr"] = np.float64(opt.param_groups[0]["lr"])
        out = model(x)
        loss = loss_fn(out, y)
        d["train/logits"] = _np(out)
        d["train/ce"] = _np(loss_fn.get_ce_loss(out, y))
        d["train/dice"] = _np(loss_fn.get_dice_loss(out, y))
        d["train/loss"] = _np(loss)
        opt.zero_grad()
        loss.backward()
        for n, p in model.named_parameters():
            d["grad/" + n] = _np(p.grad)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10.0)
        d["train/grad_norm"] = _np(gn)
        opt.step()
        for k, v in model.state_dict().items():
            d["post/" + k] = _np(v)
    np.savez_compressed(os.path.join(OUT, f"unet_{tag}.npz"), **d)
    print(tag, sum(v.nbytes for v in d.values()) // 1024, "KiB raw")


def gen_losses(ref):
    gen = torch.Generator().manual_seed(11)
    b, k1, h, w = 3, 4, 16, 20
    logits = torch.randn(b, k1, h, w, generator=gen) * 2
    labels = _labels(gen, b, h, w, k1)[:, :h, :w]
    d = {"logits": _np(logits), "labels": _np(labels)}
    for do_bg in (False, True):
        for batch in (False, True):
            for squared in (False, True):
                logit_in = logits.clone().requires_grad_(True)
                fn = ref.dice_loss.DiceLoss(k1 - 1, smooth=1e-5, do_bg=do_bg, softmax=True, batch=batch, squared=squared)
                v = fn(logit_in, labels)
                v.backward()
                key = f"dice_bg{int(do_bg)}_b{int(batch)}_s{int(squared)}"
                d[key] = _np(v)
                d[key + "_grad"] = _np(logit_in.grad)
    li = logits.clone().requires_grad_(True)
    ce = ref.ce_loss.RobustCrossEntropyLoss()(li, labels[:, None])
    ce.backward()
    d["ce"] = _np(ce)
    d["ce_grad"] = _np(li.grad)
    li = logits.clone().requires_grad_(True)
    comp = ref.compound.DiceAndCELoss(dice_kwargs=dict(num_classes=k1 - 1, do_bg=True))
    v = comp(li, labels, dice_weight=0.7, ce_weight=0.3)
    v.backward()
    d["dice_ce_w"] = _np(v)
    d["dice_ce_w_grad"] = _np(li.grad)
    d["dice_ce_zero_weight_quirk"] = _np(comp(logits, labels, dice_weight=0.0, ce_weight=None))
    # dense (soft) targets with the logits' shape: the reference skips its one-hot encoder (dice_loss.py:40-41)
    soft = torch.softmax(torch.randn(b, k1, h, w, generator=gen) * 3, dim=1)
    d["soft_targets"] = _np(soft)
    for squared in (False, True):
        li = logits.clone().requires_grad_(True)
        v = ref.dice_loss.DiceLoss(k1 - 1, do_bg=False, squared=squared)(li, soft)
        v.backward()
        d[f"dense_dice_s{int(squared)}"] = _np(v)
        d[f"dense_dice_s{int(squared)}_grad"] = _np(li.grad)
    li = logits.clone().requires_grad_(True)
    comp_t = ref.compound.DiceAndCELoss(dice_kwargs=dict(num_classes=k1 - 1, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    v = comp_t(li, soft)
    v.backward()
    d["dense_dice_ce"] = _np(v)
    d["dense_dice_ce_grad"] = _np(li.grad)
    # known-answer tests
    lab = torch.tensor([[[0, 1], [2, 2]]])
    uni = torch.zeros(1, 3, 2, 2)
    fn = ref.dice_loss.DiceLoss(2, do_bg=True)
    d["kat_uniform_dice"] = _np(fn(uni, lab))
    perfect = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float() * 100.0
    d["kat_perfect_dice"] = _np(fn(perfect, lab))
    d["kat_uniform_ce"] = _np(torch.nn.CrossEntropyLoss()(uni, lab))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **d)


def gen_poly(ref):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.1)
    rows = []
    for (lr0, n, w, interval) in [(1e-3, 4000, 250, 1), (1e-2, 100, 10, 1), (5e-4, 1000, 100, 4)]:
        s = ref.lr_scheduler.PolyLRScheduler(opt, lr0, n, w, interval=interval)
        for it in [0, 1, w - 1, w, w + 1, n // 2, n - 1]:
            s.step(it)
            rows.append([lr0, n, w, interval, it, opt.param_groups[0]["lr"]])
    np.savez_compressed(os.path.join(OUT, "poly_lr.npz"), table=np.array(rows, dtype=np.float64))


def gen_transforms(ref):
    gen = torch.Generator().manual_seed(3)
    img = torch.rand(1, 24, 40, generator=gen)
    lab = _labels(gen, 1, 24, 40, 3)
    d = {"image": _np(img), "label": _np(lab)}

    def run(t, seed, key):
        torch.manual_seed(seed)
        out = t({"image": img.clone(), "label": lab.clone()})
        d[key + "/image"] = _np(out["image"])
        d[key + "/label"] = _np(out["label"])

    # replay the draws with the same seed to record the parameters
    torch.manual_seed(21); g = torch.rand(1) * (1.5 - 0.7) + 0.7
    d["gamma/gamma"] = _np(g)
    run(ref.t_image.RandomGamma((0.7, 1.5)), 21, "gamma")

    torch.manual_seed(22); sigma = torch.rand(1).item() * 0.1; noise = torch.normal(0, sigma, size=img.shape)
    d["noise/sigma"] = np.float64(sigma); d["noise/noise"] = _np(noise)
    run(ref.t_image.RandomGaussianNoise((0, 0.1)), 22, "noise")

    torch.manual_seed(23); sc = (torch.rand(2) * 0.5 + 0.5).tolist()
    d["lowres/scales"] = np.array(sc, dtype=np.float64)
    run(ref.t_image.SimulateLowRes((0.5, 1)), 23, "lowres")

    torch.manual_seed(24); k = int(torch.randint(0, 4, (1,)).item())
    d["rot90/k"] = np.int64(k)
    run(ref.t_joint.RandomRotation90(), 24, "rot90")
    run(ref.t_joint.MirrorTransform((-1,)), 25, "mirror_w")
    run(ref.t_joint.MirrorTransform((-2, -1)), 25, "mirror_hw")
    run(ref.t_norm.ZScoreNormalize(), 26, "zscore")

    # combinator draw order: RandomTransform draws one uniform before the inner transform
    comp = ref.t_common.ComposeTransform([
        ref.t_common.RandomTransform(ref.t_image.RandomGamma((0.7, 1.5)), p=0.5),
        ref.t_common.RandomTransform(ref.t_image.SimulateLowRes((0.5, 1)), p=0.5),
        ref.t_common.RandomTransform(ref.t_joint.RandomRotation90(), p=0.5),
        ref.t_common.RandomTransform(ref.t_image.RandomGamma((0.7, 1.5)), p=0.5),
    ])
    for seed in (100, 101, 102, 103):
        run(comp, seed, f"compose_{seed}")
    d["compose/params_json"] = np.array(str(comp.get_params_dict()))
    d["blur_ksize_table"] = np.array(
        [[s, ref.t_image.RandomGaussianBlur((0.5, 1.0))._get_kernel_size(s)] for s in (0.5, 0.6, 0.62, 0.63, 0.75, 0.9, 1.0)],
        dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **d)


class _SelDS(torch.utils.data.Dataset):
    """List-of-dicts dataset with the `image_idx` member the reference selectors read."""

    def __init__(self, images, names):
        self.images, self.image_idx = images, list(names)

    def __len__(self):
        return len(self.image_idx)

    def __getitem__(self, i):
        return {"image": self.images[i], "case_name": self.image_idx[i]}


class _SelActive:
    """Duck-typed stand-in for datasets/active_dataset.py:ActiveDataset (the four members the selectors touch)."""

    def __init__(self, images, n_labeled):
        names = [f"case_{i:02d}" for i in range(len(images))]
        self.labeled_dataset = _SelDS(images[:n_labeled], names[:n_labeled])
        self.pool_dataset = _SelDS(images[n_labeled:], names[n_labeled:])

    def get_size(self):
        return len(self.labeled_dataset), len(self.pool_dataset)

    def get_pool_dataset(self):
        return self.pool_dataset

    def get_train_dataset(self):
        return self.labeled_dataset


def gen_selectors(ref):
    """the reference's own selector classes (entropy / confidence / margin / coreset / k-means / BADGE) driven
    by the reference UNet on a synthetic pool; `kcenter_greedy` on a fixed distance matrix."""
    al = ref.load_selectors()
    torch.manual_seed(7)
    model = ref.unet.UNet(2, 1, 3, [8, 16, 32], normalization="instance", dropout_prob=None)
    g = torch.Generator().manual_seed(9)
    images = torch.rand(14, 1, 32, 32, generator=g)
    images *= torch.linspace(0.2, 3.0, 14).view(-1, 1, 1, 1)  # spread the uncertainty: no near-tie rankings
    n_labeled = 4
    ad = _SelActive(images, n_labeled)
    cpu = torch.device("cpu")
    d = {"images": _np(images), "n_labeled": np.int64(n_labeled)}
    for k, v in model.state_dict().items():
        d["init/" + k] = _np(v)
    model.eval()
    with torch.no_grad():
        d["pool_logits"] = _np(model(images[n_labeled:]))
        d["enc_feature"] = _np(model.get_enc_feature(images))
    for short, cls in (("entropy", al.entropy.EntropySelector), ("confidence", al.confidence.ConfidenceSelector),
                       ("margin", al.margin.MarginSelector)):
        sel = cls(batch_size=4, num_workers=0, pin_memory=False)
        scores, names = sel.cal_scores(ad, model, cpu)
        d[f"{short}/scores"] = _np(torch.stack(scores))
        d[f"{short}/names"] = np.array(names)
        d[f"{short}/picks3"] = np.array(sel.select_next_batch(ad, 3, model, cpu))
    for metric, crit in (("cosine", "min"), ("l2", "min"), ("l2", "mean")):
        cs = al.coreset.CoresetSelector(batch_size=5, num_workers=0, pin_memory=False, metric=metric, coreset_criteria=crit)
        core, all_list, _, feats, dist = cs.cal_scores(ad, model, cpu)
        key = f"coreset_{metric}_{crit}"
        d[key + "/core"], d[key + "/all"], d[key + "/feats"], d[key + "/dist"] = core, all_list, feats, dist
        d[key + "/picks4"] = np.array(cs.select_next_batch(ad, 4, model, cpu))
    rg = np.random.default_rng(5)
    pts = rg.normal(size=(24, 6))
    dm = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    d["kcenter/dist"] = dm
    d["kcenter/init"] = np.array([0, 5, 9])
    for crit in ("min", "mean"):
        d[f"kcenter/{crit}_b6"] = np.array(sorted(int(i) for i in al.coreset.kcenter_greedy(dm, 24, 6, [0, 5, 9], crit)))
    km = al.kmean.KMeanSelector(batch_size=5, num_workers=0, pin_memory=False, metric="l2")
    lf, pf, ln, pn, p2l = km.cal_scores(ad, model, cpu)
    d["kmean/labeled_feats"], d["kmean/pool_feats"], d["kmean/pool2labeled"] = lf, pf, p2l
    d["kmean/pool_names"] = pn
    np.random.seed(0)
    d["kmean/picks3_npseed0"] = np.array(sorted(km.select_next_batch(ad, 3, model, cpu)))
    comp = ref.compound.DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    bd = al.badge.BADGESelector(dice_loss=comp.dice_loss, ce_loss=comp.ce_loss, batch_size=1, num_workers=0, pin_memory=False)
    bn, be = bd.cal_scores(ad, model, cpu)
    d["badge/names"], d["badge/embeds"] = bn, be
    # labelled set empty -> random pick (entropy_selector.py:62-70)
    ad0 = _SelActive(images, 0)
    torch.manual_seed(5)
    d["entropy/picks5_empty_seed5"] = np.array(al.entropy.EntropySelector(4, 0, False).select_next_batch(ad0, 5, model, cpu))
    np.savez_compressed(os.path.join(OUT, "selectors.npz"), **d)
    print("selectors", sum(np.asarray(v).nbytes for v in d.values()) // 1024, "KiB raw")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    ref = _refload.Ref()
    gen_unet(ref, "instance", "instance", [4, 8, 16], 32)
    gen_unet(ref, "batch", "batch", [4, 8, 16], 32)
    gen_unet(ref, "instance_odd", "instance", [6, 10, 20], 24, batch=3, k1=4)
    gen_unet(ref, "ds", "instance", [4, 8, 16, 32], 32, deep_supervision=True, ds_layer=3, train_step=False)
    gen_unet(ref, "res", "instance", [4, 8, 16], 32, block_type="res")
    gen_losses(ref)
    gen_poly(ref)
    gen_transforms(ref)
    gen_selectors(ref)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden total", tot // 1024, "KiB")


if __name__ == "__main__":
    main()
