"""The oracle (CPU restatement) against vectors produced by the imported reference
(oracle/gen_golden.py).  Runs on CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import losses_ref, train_ref, transforms_ref as T, unet_ref

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


def params_from(d, prefix="init/"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in d.items() if k.startswith(prefix)}


@pytest.mark.parametrize("tag,norm,k1", [("instance", "instance", 3), ("batch", "batch", 3), ("instance_odd", "instance", 4)])
def test_unet_train_step(tag, norm, k1):
    d = load(f"unet_{tag}.npz")
    p = params_from(d)
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["labels"])
    with torch.no_grad():
        pe = {k: v.clone() for k, v in p.items()}
        lo = unet_ref.unet_forward(pe, x, norm, training=False)
        np.testing.assert_allclose(lo.numpy(), d["eval/logits"], atol=1e-6)
        assert (lo.softmax(1).argmax(1).numpy() == d["eval/argmax"]).all()
        np.testing.assert_allclose(unet_ref.enc_feature(pe, x, norm).numpy(), d["eval/enc_feature"], atol=1e-6)
        np.testing.assert_allclose(unet_ref.pixel_feature(pe, x, norm)[1].numpy(), d["eval/pixel_feature"], atol=1e-6)
    opt = train_ref.make_optimizer(p, "adamw", weight_decay=5e-4)
    lr = train_ref.poly_lr(0, 1e-3, 4000, 250)
    assert lr == float(d["train/lr"])
    r = train_ref.train_step(p, opt, x, y, k1 - 1, norm, lr=lr)
    np.testing.assert_allclose(r["logits"].numpy(), d["train/logits"], atol=1e-6)
    np.testing.assert_allclose(r["loss"].numpy(), d["train/loss"], atol=1e-6)
    np.testing.assert_allclose(r["grad_norm"].numpy(), d["train/grad_norm"], rtol=1e-5)
    pinned = total = 0
    for k, v in d.items():
        if k.startswith("post/"):
            got = p[k[5:]].detach().numpy()
            # Adam's first step is lr*g/(|g|+eps) ~ +-lr: where the reference gradient is rounding noise (conv bias in front of a norm
            # layer: analytically 0) only |delta| <= 2 lr holds (a sign flip); that blanket bound pins nothing by itself -- the masked
            # check below does (a skipped optimizer step would be off by lr = 4e-6 there, a wrong sign by 2 lr)
            np.testing.assert_allclose(got, v, atol=2.02 * lr, err_msg=k)
            g = d.get("grad/" + k[5:])
            if g is not None and v.dtype.kind == "f":
                m = np.abs(g) > 1e-5
                np.testing.assert_allclose(got[m], v[m], atol=2e-7, err_msg=k)
                pinned += int(m.sum()); total += m.size
    assert pinned > 0.5 * total, (pinned, total)


def test_unet_grads_match():
    d = load("unet_instance.npz")
    p = params_from(d)
    for v in train_ref.trainable(p).values():
        v.requires_grad_(True)
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["labels"])
    out = unet_ref.unet_forward(p, x, "instance", True)
    np.testing.assert_allclose(losses_ref.ce_loss(out, y).item(), d["train/ce"], atol=1e-6)
    np.testing.assert_allclose(losses_ref.dice_loss(out, y, 2, do_bg=True).item(), d["train/dice"], atol=1e-6)
    losses_ref.dice_and_ce(out, y, 2).backward()
    for k, v in d.items():
        if k.startswith("grad/"):
            np.testing.assert_allclose(p[k[5:]].grad.numpy(), v, atol=2e-6, err_msg=k)


def test_deep_supervision_and_res():
    d = load("unet_ds.npz")
    p = params_from(d)
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        outs = unet_ref.unet_forward(p, x, "instance", False, deep_supervision=True, ds_layer=3, return_ds=True)
    assert len(outs) == 3
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), d[f"eval/ds{i}"], atol=1e-6)
    d = load("unet_res.npz")
    p = params_from(d)
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        lo = unet_ref.unet_forward(p, x, "instance", False, block_type="res")
    np.testing.assert_allclose(lo.numpy(), d["eval/logits"], atol=1e-6)


def test_init_matches_reference_rng_stream():
    d = load("unet_instance.npz")
    # gen_golden perturbs norm affine params after init; conv weights are untouched
    p = unet_ref.init_params(1, 3, [4, 8, 16], "instance", seed=1337)
    for k in ("encoder.levels.0.0.all.0.weight", "encoder.levels.2.1.all.0.bias",
              "decoder.upsamples.0.weight", "decoder.upsamples.1.bias", "decoder.seg_output.weight"):
        np.testing.assert_array_equal(p[k].numpy(), d["init/" + k], err_msg=k)
    assert set(p) == {k[5:] for k in d if k.startswith("init/")}


def test_losses():
    d = load("losses.npz")
    logits, labels = torch.from_numpy(d["logits"]), torch.from_numpy(d["labels"])
    for do_bg in (False, True):
        for batch in (False, True):
            for squared in (False, True):
                key = f"dice_bg{int(do_bg)}_b{int(batch)}_s{int(squared)}"
                li = logits.clone().requires_grad_(True)
                v = losses_ref.dice_loss(li, labels, 3, do_bg=do_bg, batch=batch, squared=squared)
                v.backward()
                np.testing.assert_allclose(v.item(), d[key], atol=1e-6)
                np.testing.assert_allclose(li.grad.numpy(), d[key + "_grad"], atol=1e-7)
    np.testing.assert_allclose(losses_ref.ce_loss(logits, labels[:, None]).item(), d["ce"], atol=1e-6)
    np.testing.assert_allclose(losses_ref.dice_and_ce(logits, labels, 3, 0.7, 0.3).item(), d["dice_ce_w"], atol=1e-6)
    np.testing.assert_allclose(losses_ref.dice_and_ce(logits, labels, 3, 0.0, None).item(),
                               d["dice_ce_zero_weight_quirk"], atol=1e-6)
    soft = torch.from_numpy(d["soft_targets"])
    for squared in (False, True):
        li = logits.clone().requires_grad_(True)
        v = losses_ref.dice_loss(li, soft, 3, do_bg=False, squared=squared)
        v.backward()
        np.testing.assert_allclose(v.item(), d[f"dense_dice_s{int(squared)}"], atol=1e-6)
        np.testing.assert_allclose(li.grad.numpy(), d[f"dense_dice_s{int(squared)}_grad"], atol=1e-7)
    np.testing.assert_allclose(losses_ref.dice_and_ce(logits, soft, 3).item(), d["dense_dice_ce"], atol=1e-6)
    # known answers
    lab = torch.tensor([[[0, 1], [2, 2]]])
    assert abs(losses_ref.dice_loss(torch.zeros(1, 3, 2, 2), lab, 2, do_bg=True).item() - 0.6761878354) < 1e-6
    assert abs(float(d["kat_uniform_dice"]) - 0.6761878354) < 1e-6
    assert abs(float(d["kat_perfect_dice"])) < 1e-6
    assert abs(float(d["kat_uniform_ce"]) - np.log(3)) < 1e-6


def test_poly_lr():
    for lr0, n, w, interval, it, want in load("poly_lr.npz")["table"]:
        got = train_ref.poly_lr(int(it), lr0, int(n), int(w), interval=int(interval))
        assert got == pytest.approx(want, rel=1e-12, abs=0)


def test_transforms_pinned():
    d = load("transforms.npz")
    img, lab = torch.from_numpy(d["image"]), torch.from_numpy(d["label"])
    np.testing.assert_array_equal(T.apply_gamma(img, torch.from_numpy(d["gamma/gamma"])).numpy(), d["gamma/image"])
    np.testing.assert_array_equal(T.apply_noise(img, torch.from_numpy(d["noise/noise"])).numpy(), d["noise/image"])
    np.testing.assert_array_equal(T.apply_lowres(img, d["lowres/scales"].tolist()).numpy(), d["lowres/image"])
    k = int(d["rot90/k"])
    np.testing.assert_array_equal(T.apply_rot90(img, k).numpy(), d["rot90/image"])
    np.testing.assert_array_equal(T.apply_rot90(lab, k).numpy(), d["rot90/label"])
    np.testing.assert_array_equal(T.apply_mirror(img, (-1,)).numpy(), d["mirror_w/image"])
    np.testing.assert_array_equal(T.apply_mirror(lab, (-2, -1)).numpy(), d["mirror_hw/label"])
    np.testing.assert_array_equal(T.apply_zscore(img).numpy(), d["zscore/image"])
    for s, k in d["blur_ksize_table"]:
        assert T.blur_kernel_size(float(s)) == int(k)


def test_transform_draw_order():
    """RandomTransform draws its uniform before the inner transform's draws (common.py:27)."""
    d = load("transforms.npz")
    img, lab = torch.from_numpy(d["image"]), torch.from_numpy(d["label"])
    for seed in (100, 101, 102, 103):
        torch.manual_seed(seed)
        im, lb = img.clone(), lab.clone()
        if T.draw_apply(0.5):
            im = T.apply_gamma(im, T.draw_gamma(0.7, 1.5))
        if T.draw_apply(0.5):
            im = T.apply_lowres(im, T.draw_lowres_scales(2, 0.5, 1.0))
        if T.draw_apply(0.5):
            k = int(torch.randint(0, 4, (1,)).item())
            im, lb = T.apply_rot90(im, k), T.apply_rot90(lb, k)
        if T.draw_apply(0.5):
            im = T.apply_gamma(im, T.draw_gamma(0.7, 1.5))
        np.testing.assert_array_equal(im.numpy(), d[f"compose_{seed}/image"])
        np.testing.assert_array_equal(lb.numpy(), d[f"compose_{seed}/label"])


def test_affine_grid_restatement_equals_torch_bmm_on_this_cpu():
    """oracle._affine_grid writes torchvision's `base.bmm(theta')` out as k-sequential fused accumulation so that it is
    machine independent; here it is pinned to torch's own bmm, bit for bit, as run on the dev container's CPU (the GPU
    kernel csrc/augment.hip::affine_src follows the same order).  Another CPU may dispatch a different sgemm kernel; a
    difference there is reported as a skip, not a failure."""
    import numpy as np
    from oracle import transforms_ref as R
    rng = np.random.default_rng(3)
    bad = tot = 0
    for t in range(40):
        h, w = int(rng.integers(30, 400)), int(rng.integers(30, 400))
        ang = float(rng.uniform(-20, 20)) if t % 3 else float(rng.integers(-20, 21))
        sc = float(rng.uniform(0.7, 1.4)) if t % 2 else 1.0
        sh = [float(rng.uniform(-8, 8)), 0.0] if t % 4 == 0 else [0.0, 0.0]
        tr = [float(rng.integers(-6, 7)), float(rng.integers(-6, 7))] if t % 5 == 0 else [0.0, 0.0]
        m = R.inverse_affine_matrix([0.0, 0.0], ang, tr, sc, sh)
        g1, g2 = R._affine_grid(m, w, h), R._affine_grid_bmm(m, w, h)
        bad += int((g1 != g2).sum())
        tot += g1.numel()
    if bad and torch.backends.cpu.get_cpu_capability() != "AVX512":
        pytest.skip(f"{bad} of {tot} grid values differ from this CPU's sgemm ({torch.backends.cpu.get_cpu_capability()})")
    assert bad == 0, (bad, tot)


def test_elastic_restatement_known_answers():
    """oracle.apply_elastic restates the build's OWN elastic-deformation spec (the reference has none):
    zero field = identity, constant integer field = shift with zero fill, label values never invented."""
    from oracle import transforms_ref as R
    g = torch.Generator().manual_seed(0)
    img = torch.rand(2, 20, 24, generator=g)
    lab = torch.randint(0, 3, (1, 20, 24), generator=g)
    assert torch.equal(R.apply_elastic(img, torch.zeros(2, 3, 4)), img)
    assert torch.equal(R.apply_elastic(lab, torch.zeros(2, 3, 4)), lab)
    d = torch.zeros(2, 2, 2)
    d[0] += 3
    d[1] -= 2
    want = torch.zeros_like(lab[0])
    want[2:, :-3] = lab[0][:-2, 3:]
    assert torch.equal(R.apply_elastic(lab, d)[0], want)
    wi = torch.zeros_like(img)
    wi[:, 2:, :-3] = img[:, :-2, 3:]
    assert torch.allclose(R.apply_elastic(img, d), wi, atol=1e-6)
    rnd = torch.randn(2, 4, 4, generator=g) * 5
    out = R.apply_elastic(lab, rnd)
    assert set(out.unique().tolist()) <= {0, 1, 2} and out.shape == lab.shape


def test_selectors_oracle_vs_reference_vectors():
    """`oracle/selectors_ref.py` against vectors the reference's own selector classes produced
    (entropy_selector.py:42-49, confidence_selector.py:42-47, margin_selector.py:42-48, coreset_selector.py:19-52,
    kmean_selector.py:95-104, badge_selector.py:19-34), driven by the reference UNet in `oracle/gen_golden.gen_selectors`."""
    from oracle import selectors_ref as S
    d = load("selectors.npz")
    p = params_from(d)
    images = torch.from_numpy(d["images"])
    nl = int(d["n_labeled"])
    with torch.no_grad():
        logits = unet_ref.unet_forward(p, images[nl:], "instance", training=False)
        feats = unet_ref.enc_feature(p, images, "instance").numpy()
    np.testing.assert_allclose(logits.numpy(), d["pool_logits"], atol=1e-6)
    np.testing.assert_allclose(feats, d["enc_feature"], atol=1e-6)
    ref_logits = torch.from_numpy(d["pool_logits"])
    names = [f"case_{i:02d}" for i in range(len(images))]
    for short, fn in (("entropy", S.entropy_score), ("confidence", S.confidence_score), ("margin", S.margin_score)):
        got = fn(ref_logits)
        np.testing.assert_allclose(got.numpy(), d[f"{short}/scores"], rtol=1e-6, atol=1e-7, err_msg=short)
        assert list(d[f"{short}/names"]) == names[nl:]
        order = torch.sort(got, descending=True)[1][:3]
        assert [names[nl + int(i)] for i in order] == list(d[f"{short}/picks3"]), short
    for crit in ("min", "mean"):
        got = S.kcenter_greedy(d["kcenter/dist"], 24, 6, d["kcenter/init"].tolist(), crit)
        assert sorted(int(i) for i in got) == d[f"kcenter/{crit}_b6"].tolist(), crit
    from sklearn.metrics import pairwise_distances
    for metric, crit in (("cosine", "min"), ("l2", "min"), ("l2", "mean")):
        key = f"coreset_{metric}_{crit}"
        np.testing.assert_allclose(d[key + "/feats"], d["enc_feature"], atol=0)
        dm = pairwise_distances(d["enc_feature"], metric=metric)
        np.testing.assert_allclose(dm / dm.sum(), d[key + "/dist"], rtol=1e-5, atol=1e-9)
        got = S.kcenter_greedy(d[key + "/dist"], len(images), 4, np.arange(nl), crit)
        assert sorted(names[int(i)] for i in got) == sorted(d[key + "/picks4"].tolist()), key
    np.testing.assert_allclose(S.row_standardise(d["enc_feature"][nl:]), d["kmean/pool_feats"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(S.row_standardise(d["enc_feature"][:nl]), d["kmean/labeled_feats"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pairwise_distances(d["kmean/pool_feats"], d["kmean/labeled_feats"], metric="l2"),
                               d["kmean/pool2labeled"], rtol=1e-5, atol=1e-6)
    # BADGE gradient embedding = d(CE + Dice on the pseudo labels) / d(decoder.seg_output.weight), one image per batch
    for i in range(len(images) - nl):
        q = {k: v.clone().requires_grad_(k == "decoder.seg_output.weight") for k, v in p.items()}
        out = unet_ref.unet_forward(q, images[nl + i:nl + i + 1], "instance", training=False)
        pred = out.softmax(1).argmax(1)
        loss = losses_ref.ce_loss(out, pred) + losses_ref.dice_loss(out, pred, 2, do_bg=True)
        (gr,) = torch.autograd.grad(loss, q["decoder.seg_output.weight"])
        np.testing.assert_allclose(gr.flatten().numpy(), d["badge/embeds"][i], rtol=1e-4, atol=1e-7)
    torch.manual_seed(5)
    idx = torch.sort(torch.rand(len(images)), descending=True)[1][:5]
    assert [names[int(i)] for i in idx] == list(d["entropy/picks5_empty_seed5"])
