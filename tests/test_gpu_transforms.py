"""Augmentation / resize / z-score kernels against the oracle (oracle/transforms_ref.py) with the SAME drawn
parameters, and against vectors produced by the reference's own transforms (tests/golden/transforms.npz)
where the reference is runnable (gamma, noise, low-res, rot90, mirror, z-score).  The torchvision-backed
transforms (affine, rotation, blur, contrast, resize) are PARITY UNPINNED: they are checked against the
oracle's restatement of torchvision's algorithm plus analytic known answers."""
import os

import numpy as np
import pytest
import torch

from oracle import transforms_ref as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _sample(h=36, w=52, c=1, seed=0, k1=3):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(c, h, w, generator=g)
    lab = torch.randint(0, k1, (1, h // 4 + 1, w // 4 + 1), generator=g).float()
    lab = torch.nn.functional.interpolate(lab[None], size=(h, w), mode="nearest")[0].long()
    return img, lab


def test_golden_pure_torch_transforms():
    from transforms import functional_hip as FH
    dev = _dev()
    d = dict(np.load(os.path.join(G, "transforms.npz")))
    img = torch.from_numpy(d["image"]).to(dev)[None]
    lab = torch.from_numpy(d["label"]).to(dev)
    np.testing.assert_allclose(FH.elementwise(img, FH.EW_GAMMA, p0=[float(d["gamma/gamma"][0])])[0].cpu().numpy(), d["gamma/image"], atol=2e-6)
    out = FH.elementwise(img, FH.EW_NOISE, aux=torch.from_numpy(d["noise/noise"]).to(dev)[None])
    np.testing.assert_array_equal(out[0].cpu().numpy(), d["noise/image"])
    sc = d["lowres/scales"].tolist()
    low = [int(s * i) for s, i in zip(sc, img.shape[-2:])]
    np.testing.assert_allclose(FH.lowres(img, [low])[0].cpu().numpy(), d["lowres/image"], atol=1e-6)
    k = int(d["rot90/k"])
    np.testing.assert_array_equal(FH.rot90_flip(img, k)[0].cpu().numpy(), d["rot90/image"])
    np.testing.assert_array_equal(FH.rot90_flip(lab, k).cpu().numpy(), d["rot90/label"])
    np.testing.assert_array_equal(FH.rot90_flip(img, 0, False, True)[0].cpu().numpy(), d["mirror_w/image"])
    np.testing.assert_array_equal(FH.rot90_flip(lab, 0, True, True).cpu().numpy(), d["mirror_hw/label"])
    ms = FH.sample_stats(img)
    np.testing.assert_allclose(FH.elementwise(img, FH.EW_ZSCORE, mean_std=ms)[0].cpu().numpy(), d["zscore/image"], atol=2e-5)


@pytest.mark.parametrize("params", [(0.0, (0, 0), 1.0, (0.0, 0.0)), (12.5, (0, 0), 1.0, (0.0, 0.0)), (0.0, (0, 0), 0.8, (0.0, 0.0)),
                                    (-7.0, (3, -2), 1.3, (4.0, 0.0)), (90.0, (0, 0), 1.0, (0.0, 0.0)), (0.0, (0, 0), 2.0, (0.0, 0.0))])
def test_affine_matches_oracle_and_known_answers(params):
    from transforms import functional_hip as FH
    from transforms.joint_transform import inverse_affine_matrix
    dev = _dev()
    angle, tr, sc, sh = params
    for hw in ((36, 52), (40, 40)):
        img, lab = _sample(*hw, c=1, seed=3)
        m = inverse_affine_matrix([0.0, 0.0], angle, [1.0 * t for t in tr], sc, list(sh))
        oi, ol = FH.affine_nearest(img[None].to(dev), lab.to(dev), [m])
        ri, rl = R.apply_affine(img, angle, tr, sc, sh), R.apply_affine(lab, angle, tr, sc, sh)
        # nearest sampling is index work: the kernel pins torchvision's fp32 operation order (csrc/augment.hip affine_src),
        # so image and label maps are bit-exact, .5 ties included
        assert torch.equal(oi[0].cpu(), ri)
        assert torch.equal(ol.cpu(), rl)
        if params == (0.0, (0, 0), 1.0, (0.0, 0.0)):  # identity
            assert torch.equal(oi[0].cpu(), img) and torch.equal(ol.cpu(), lab)
        if params[0] == 90.0 and hw[0] == hw[1]:  # 90 degrees on a square image == rot90 (nearest, exact)
            assert torch.equal(oi[0].cpu(), R.apply_affine(img, 90.0, (0, 0), 1.0, (0.0, 0.0)))
            assert torch.equal(oi[0].cpu(), torch.rot90(img, 1, (-2, -1))) or torch.equal(oi[0].cpu(), torch.rot90(img, -1, (-2, -1)))


def test_rotation_and_apply_flags():
    from transforms import functional_hip as FH
    from transforms.joint_transform import IDENTITY, inverse_affine_matrix
    dev = _dev()
    img, lab = _sample(seed=5)
    imgs = torch.stack([img, img * 0.5]).to(dev)
    labs = torch.cat([lab, lab]).to(dev)
    m = inverse_affine_matrix([0.0, 0.0], -17.0, [0.0, 0.0], 1.0, [0.0, 0.0])  # F.rotate(angle=17)
    oi, ol = FH.affine_nearest(imgs, labs, [m, IDENTITY], [True, False])
    assert torch.equal(oi[0].cpu(), R.apply_rotate(img, 17.0)) and torch.equal(ol[0].cpu(), R.apply_rotate(lab, 17.0)[0])
    assert torch.equal(oi[1].cpu(), img * 0.5) and torch.equal(ol[1].cpu(), lab[0])  # apply=0: pass-through


def test_affine_index_maps_bit_exact_random_sweep():
    """Source-index maps of random rotations / scales / shears at FUGC-like sizes equal the oracle's exactly
    (an 'index image' makes every differing source pixel visible)."""
    from transforms import functional_hip as FH
    from transforms.joint_transform import inverse_affine_matrix
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    for t in range(24):
        h, w = [(336, 544), (128, 96), (61, 45), (256, 256)][t % 4]
        ang = float(torch.empty(1).uniform_(-20, 20, generator=g)) if t % 3 else float(torch.randint(-20, 21, (1,), generator=g))
        sc = float(torch.empty(1).uniform_(0.7, 1.4, generator=g)) if t % 2 else 1.0
        sh = [float(torch.empty(1).uniform_(-5, 5, generator=g)), 0.0] if t % 5 == 0 else [0.0, 0.0]
        lab = (torch.arange(h * w, dtype=torch.long) + 1).reshape(1, h, w)
        m = inverse_affine_matrix([0.0, 0.0], ang, [0.0, 0.0], sc, sh)
        _, ol = FH.affine_nearest(None, lab.to(dev), [m])
        assert torch.equal(ol.cpu(), R.apply_affine(lab, ang, (0, 0), sc, sh)), (t, h, w, ang, sc, sh)


def test_elastic_deformation_own_spec():
    """Elastic deformation has NO reference counterpart (parity unpinned by construction): the kernel is
    checked against the CPU restatement of the build's own spec (oracle/transforms_ref.py::apply_elastic) -- label maps
    bit-exact, images to 1e-6 -- and against known answers: zero displacement = identity, a constant integer displacement =
    a shift with zero fill, apply=0 = pass-through, per-sample draw order and determinism of the transform class."""
    from transforms import functional_hip as FH
    from transforms.gpu_pipeline import BatchedAugment, al_train_transforms
    from transforms.joint_transform import RandomElastic
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    for (h, w, c, gh, gw) in ((48, 64, 1, 4, 4), (37, 53, 3, 3, 5), (336, 544, 1, 4, 4)):
        img, lab = _sample(h, w, c=c, seed=h)
        disp = torch.randn(2, 2, gh, gw, generator=g) * torch.tensor([6.0, 0.0]).view(2, 1, 1, 1)  # sample 1: zero field
        imgs, labs = torch.stack([img, img]).to(dev), torch.cat([lab, lab]).to(dev)
        oi, ol = FH.elastic_warp(imgs, labs, disp.to(dev))
        ri, rl = R.apply_elastic(img, disp[0]), R.apply_elastic(lab, disp[0])
        assert torch.equal(ol[0].cpu(), rl[0]), (h, w)
        np.testing.assert_allclose(oi[0].cpu().numpy(), ri.numpy(), atol=1e-6)
        assert torch.equal(oi[1].cpu(), img) and torch.equal(ol[1].cpu(), lab[0])  # zero displacement = identity
        assert (ol[0].cpu() != lab[0]).float().mean().item() > 0.01  # and a 6-pixel field does move things
    # constant displacement (+3, -2): out[y][x] = in[y - 2][x + 3], zero outside
    img, lab = _sample(40, 56, c=1, seed=1)
    disp = torch.zeros(1, 2, 2, 2)
    disp[0, 0] += 3.0
    disp[0, 1] -= 2.0
    oi, ol = FH.elastic_warp(img[None].to(dev), lab.to(dev), disp.to(dev))
    want = torch.zeros_like(lab[0])
    want[2:, :-3] = lab[0][:-2, 3:]
    assert torch.equal(ol[0].cpu(), want)
    wi = torch.zeros_like(img[0])
    wi[2:, :-3] = img[0][:-2, 3:]
    np.testing.assert_allclose(oi[0, 0].cpu().numpy(), wi.numpy(), atol=1e-6)
    o2, l2 = FH.elastic_warp(img[None].to(dev), lab.to(dev), disp.to(dev), [False])
    assert torch.equal(o2[0].cpu(), img) and torch.equal(l2.cpu(), lab)
    # transform class: one torch.rand(1) then one torch.randn(2, gh, gw); reproducible under the seed; dict API
    t = RandomElastic(sigma=(2.0, 6.0), grid=(3, 4))
    torch.manual_seed(9)
    s = float(torch.rand(1).item() * 4.0 + 2.0)
    d = torch.randn(2, 3, 4) * s
    torch.manual_seed(9)
    out = t({"image": img.to(dev), "label": lab.to(dev), "case_name": "e"})
    assert out["case_name"] == "e" and out["label"].shape == (1, 40, 56) and out["label"].dtype == torch.long
    assert torch.equal(out["label"].cpu(), R.apply_elastic(lab, d))
    np.testing.assert_allclose(out["image"].cpu().numpy(), R.apply_elastic(img, d).numpy(), atol=1e-6)
    assert t.get_params_dict() == {"RandomElastic": {"sigma": [2.0, 6.0], "grid": [3, 4]}}
    # off by default in the al_train pipelines; one extra leading stage when asked for
    assert len(al_train_transforms("fugc", elastic=True).transforms) == len(al_train_transforms("fugc").transforms) + 1
    torch.manual_seed(3)
    b = BatchedAugment(al_train_transforms("busi", elastic=True), image_size=32)(torch.stack([img] * 4).to(dev), torch.cat([lab] * 4).to(dev))
    assert b["image"].shape == (4, 1, 32, 32) and b["label"].shape == (4, 32, 32)


def test_random_crop2d():
    """RandomCrop2D (joint_transform.py:130-155): T.RandomCrop.get_params draw order (i then j, none when the size already
    matches) and F.crop, per-sample dict API and batched, image + label bit-exact vs the oracle's draw + slice."""
    from transforms.joint_transform import RandomCrop2D
    dev = _dev()
    img, lab = _sample(40, 56, c=3, seed=9)
    t = RandomCrop2D((24, 32))
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        i, j, h, w = R.draw_crop(40, 56, 24, 32)
        torch.manual_seed(seed)
        out = t({"image": img.to(dev), "label": lab.to(dev), "case_name": "c"})
        assert out["image"].shape == (3, 24, 32) and out["label"].shape == (1, 24, 32) and out["case_name"] == "c"
        assert torch.equal(out["image"].cpu(), R.apply_crop(img, i, j, h, w))
        assert torch.equal(out["label"].cpu(), R.apply_crop(lab, i, j, h, w))
    # batched: per-sample windows in one launch
    imgs = torch.stack([img, img.flip(-1), img * 0.5]).to(dev)
    labs = torch.cat([lab, lab.flip(-1), lab]).to(dev)
    params = [(0, 0, 24, 32), (16, 24, 24, 32), (7, 3, 24, 32)]
    oi, ol = t.apply_batch(imgs, labs, params)
    for b, (i, j, h, w) in enumerate(params):
        assert torch.equal(oi[b], imgs[b, :, i:i + h, j:j + w]) and torch.equal(ol[b], labs[b, i:i + h, j:j + w])
    same = RandomCrop2D(40)  # int -> square; same size as the image -> (0, 0, h, w) without touching the RNG
    torch.manual_seed(5)
    before = torch.rand(1).item()
    torch.manual_seed(5)
    sq, _ = _sample(40, 40, c=1, seed=2)
    out = same({"image": sq.to(dev), "label": torch.zeros(1, 40, 40, dtype=torch.long, device=dev)})
    assert torch.equal(out["image"].cpu(), sq) and torch.rand(1).item() == before
    with pytest.raises(ValueError):
        RandomCrop2D(64)({"image": img.to(dev), "label": lab.to(dev)})
    assert t.get_params_dict() == {"RandomCrop2D": {"crop": (24, 32)}}


@pytest.mark.parametrize("sigma", [0.5, 0.62, 0.75, 1.0, 1.9])
def test_gaussian_blur(sigma):
    from transforms import functional_hip as FH
    dev = _dev()
    k = R.blur_kernel_size(sigma)
    for c in (1, 3):
        img, _ = _sample(c=c, seed=7)
        out = FH.gaussian_blur(img[None].to(dev), [sigma], [k])[0].cpu()
        np.testing.assert_allclose(out.numpy(), R.apply_gaussian_blur(img, k, sigma).numpy(), atol=2e-6)
    const = torch.full((1, 1, 20, 24), 0.37, device=dev)
    np.testing.assert_allclose(FH.gaussian_blur(const, [sigma], [k]).cpu().numpy(), 0.37, atol=1e-6)  # kernel sums to 1, reflect pad


@pytest.mark.parametrize("c", [1, 3])
def test_contrast_gamma_zscore(c):
    from transforms import functional_hip as FH
    dev = _dev()
    img, _ = _sample(c=c, seed=9)
    x = img[None].to(dev)
    ms = FH.sample_stats(x, gray=(c == 3))
    for f in (1.0, 0.0, 0.8, 1.25):
        out = FH.elementwise(x, FH.EW_CONTRAST, p0=[f], mean_std=ms)[0].cpu()
        np.testing.assert_allclose(out.numpy(), R.apply_contrast(img, f).numpy(), atol=2e-6)
    np.testing.assert_allclose(FH.elementwise(x, FH.EW_CONTRAST, p0=[1.0], mean_std=ms)[0].cpu().numpy(), img.numpy(), atol=1e-7)
    np.testing.assert_allclose(FH.elementwise(x, FH.EW_GAMMA, p0=[1.0])[0].cpu().numpy(), img.numpy(), atol=1e-6)
    np.testing.assert_allclose(FH.elementwise(x, FH.EW_GAMMA, p0=[1.37])[0].cpu().numpy(), R.apply_gamma(img, 1.37).numpy(), atol=2e-6)
    z = FH.elementwise(x, FH.EW_ZSCORE, mean_std=FH.sample_stats(x))[0].cpu()
    np.testing.assert_allclose(z.numpy(), R.apply_zscore(img).numpy(), atol=2e-5)
    assert abs(z.mean().item()) < 1e-5 and abs(z.std().item() - 1.0) < 1e-4


@pytest.mark.parametrize("size", [(24, 24), (36, 52), (72, 80), (17, 23)])
def test_resize(size):
    from transforms import functional_hip as FH
    dev = _dev()
    img, lab = _sample(seed=11)
    x, l = img[None].to(dev), lab.to(dev)
    for aa in (False, True):
        out = FH.resize_bilinear(x, size[0], size[1], antialias=aa)[0].cpu()
        np.testing.assert_allclose(out.numpy(), R.apply_resize_image(img, size, antialias=aa).numpy(), atol=3e-6)
    assert torch.equal(FH.resize_nearest(l, size[0], size[1]).cpu(), R.apply_resize_label(lab, size))
    assert torch.equal(FH.resize_bilinear(x, 36, 52)[0].cpu(), img)  # same size == identity
    for scales in ([0.5, 0.5], [0.93, 0.61], [1.0, 1.0]):
        low = [int(s * i) for s, i in zip(scales, img.shape[1:])]
        np.testing.assert_allclose(FH.lowres(x, [low])[0].cpu().numpy(), R.apply_lowres(img, scales).numpy(), atol=2e-6)


def test_noise_kernel_statistics():
    from transforms import functional_hip as FH
    dev = _dev()
    x = torch.full((2, 1, 256, 256), 0.5, device=dev)
    out = FH.noise_clip(x, [0.05, 0.1], seed=1234, offset=1, apply=[True, False])
    n = (out[0] - 0.5).flatten()
    assert abs(n.mean().item()) < 1e-3 and abs(n.std().item() - 0.05) < 2e-3
    assert torch.equal(out[1], x[1])
    out2 = FH.noise_clip(x, [0.05, 0.1], seed=1234, offset=1, apply=[True, False])
    assert torch.equal(out, out2)  # counter-based: reproducible
    assert not torch.equal(out, FH.noise_clip(x, [0.05, 0.1], seed=1234, offset=2, apply=[True, False]))


def test_dropin_transform_classes_and_batched_pipeline():
    """Per-sample dict API (reference contract) == batched pipeline with the same RNG seed; draw order follows
    the reference (RandomTransform draws one uniform before the inner transform)."""
    from transforms.common import ComposeTransform, RandomTransform
    from transforms.gpu_pipeline import BatchedAugment, al_train_transforms
    from transforms.image_transform import RandomGamma, RandomGaussianNoise, SimulateLowRes
    from transforms.joint_transform import JointResize, RandomRotation90
    from transforms.normalization import ZScoreNormalize
    dev = _dev()
    d = dict(np.load(os.path.join(G, "transforms.npz")))
    img, lab = torch.from_numpy(d["image"]).to(dev), torch.from_numpy(d["label"]).to(dev)
    comp = ComposeTransform([RandomTransform(RandomGamma((0.7, 1.5)), p=0.5), RandomTransform(SimulateLowRes((0.5, 1)), p=0.5),
                             RandomTransform(RandomRotation90(), p=0.5), RandomTransform(RandomGamma((0.7, 1.5)), p=0.5)])
    for seed in (100, 101, 102, 103):
        torch.manual_seed(seed)
        out = comp({"image": img.clone(), "label": lab.clone(), "case_name": "x"})
        assert out["case_name"] == "x" and out["label"].dtype == torch.long and out["label"].shape[0] == 1
        np.testing.assert_allclose(out["image"].cpu().numpy(), d[f"compose_{seed}/image"], atol=3e-6)
        np.testing.assert_array_equal(out["label"].cpu().numpy(), d[f"compose_{seed}/label"])
    assert "ComposeTransform" in comp.get_params_dict()
    # full al_train pipeline: per-sample calls vs one batched call, same seed
    RandomGaussianNoise.exact_rng = True
    try:
        pipe = al_train_transforms("fugc")
        g = torch.Generator().manual_seed(2)
        imgs = torch.rand(6, 1, 48, 64, generator=g).to(dev)
        labs = torch.randint(0, 3, (6, 1, 48, 64), generator=g).to(dev)
        torch.manual_seed(77)
        singles = [pipe({"image": imgs[i].clone(), "label": labs[i].clone()}) for i in range(6)]
        torch.manual_seed(77)
        batched = BatchedAugment(pipe, image_size=32, do_normalize=True)(imgs, labs)
        fin, zs = JointResize(32), ZScoreNormalize()
        for i, s in enumerate(singles):
            s = zs(fin(s))
            np.testing.assert_allclose(batched["image"][i].cpu().numpy(), s["image"].cpu().numpy(), atol=2e-5)
            assert torch.equal(batched["label"][i].cpu(), s["label"][0].cpu())
        assert batched["image"].shape == (6, 1, 32, 32) and batched["label"].shape == (6, 32, 32)
    finally:
        RandomGaussianNoise.exact_rng = False


def test_al_train_pipeline_matches_oracle_pipeline():
    """HIP transform classes vs the oracle's restatement of the al_train fugc pipeline, same seeds."""
    from transforms.gpu_pipeline import al_train_transforms
    from transforms.image_transform import RandomGaussianNoise
    dev = _dev()
    RandomGaussianNoise.exact_rng = True
    try:
        pipe = al_train_transforms("fugc")
        hits = 0
        for seed in range(40):
            img, lab = _sample(48, 64, seed=seed)
            torch.manual_seed(1000 + seed)
            ri, rl, rec = R.al_train_fugc_pipeline(img.clone(), lab.clone())
            torch.manual_seed(1000 + seed)
            out = pipe({"image": img.to(dev), "label": lab.to(dev)})
            hits += len(rec)
            oi, ol = out["image"].cpu(), out["label"].cpu()
            np.testing.assert_allclose(oi.numpy(), ri.numpy(), atol=1e-5, err_msg=str((seed, [n for n, _ in rec])))
            assert torch.equal(ol, rl), (seed, rec)  # label maps: bit-exact through the affine stages too
        assert hits > 20  # the seeds exercise the stages
    finally:
        RandomGaussianNoise.exact_rng = False


def test_train_engine_matches_golden_step():
    """TrainEngine (flat fused optimizer, on-device clip) reproduces the reference's post-step state."""
    from losses.compound_losses import DiceAndCELoss
    from losses.dice_loss import DiceLoss
    from models.unet import UNet
    from training.engine import TrainEngine
    dev = _dev()
    for tag, norm in (("instance", "instance"), ("batch", "batch")):
        d = dict(np.load(os.path.join(G, f"unet_{tag}.npz")))
        m = UNet(2, 1, 3, [4, 8, 16], normalization=norm, dropout_prob=None)
        m.load_state_dict({k[5:]: torch.from_numpy(v.copy()) for k, v in d.items() if k.startswith("init/")})
        m = m.to(dev)
        loss_fn = DiceAndCELoss(dice_loss=DiceLoss, dice_kwargs=dict(num_classes=2, smooth=1e-5, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs={})
        eng = TrainEngine(m, loss_fn, "adamw", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=4000, lr_warmup_iter=250)
        loss = eng.train_step({"image": torch.from_numpy(d["x"]), "label": torch.from_numpy(d["labels"])})
        assert abs(loss.item() - float(d["train/loss"])) < 1e-4
        assert abs(eng.optimizer.last_norm[0].item() - float(d["train/grad_norm"])) / float(d["train/grad_norm"]) < 1e-3
        lr = float(d["train/lr"])
        assert eng.optimizer.param_groups[0]["lr"] == lr
        pinned = total = 0
        for k, v in m.state_dict().items():
            ref, got = d["post/" + k], v.cpu().numpy()
            if ref.dtype.kind == "f":
                # Adam's first step is +-lr: the blanket bound (2 lr: a sign flip of a rounding-noise gradient) pins nothing by itself;
                # what pins the optimizer is the masked check -- where the reference gradient is clearly non-zero the state must agree
                # to 5e-7 (a skipped step would be off by lr = 4e-6, a wrong sign by 2 lr) -- and it must cover most of the model
                np.testing.assert_allclose(got, ref, atol=2.02 * lr + 1e-6, err_msg=k)
                g = d.get("grad/" + k)
                if g is not None:
                    msk = np.abs(g) > 1e-4
                    np.testing.assert_allclose(got[msk], ref[msk], atol=5e-7, err_msg=k)
                    pinned += int(msk.sum()); total += msk.size
            else:
                assert (got == ref).all(), k
        assert pinned > 0.5 * total, (pinned, total)
        # prediction path (valid_slices core)
        pred = eng.predict(torch.from_numpy(d["x"]))
        assert pred.shape == (2, 32, 32) and pred.dtype == torch.long


def test_selected_sample_mode_touches_only_the_drawn_samples():
    """Round 4: a stage drawn for k of B samples streams k samples.  Kernel level: `apply[b] < 0` leaves
    sample b of the output untouched (not read, not written) in every stage kernel, `mia_copy_selected` copies the selected
    samples only, `mia_sample_stats_sel` zeroes the rows it skips.  Pipeline level: BatchedAugment (selected-sample mode,
    element-wise stages in place, neighbourhood stages through a scratch buffer) gives exactly the outputs of the
    whole-batch path (`functional_hip.set_selective` never switched on), leaves the caller's tensors alone unless
    `inplace=True`, and `inplace=True` gives the same outputs."""
    import ctypes
    from mia_hip import call
    from mia_hip.ops import _c_float, _c_i64, _p, _stream
    from transforms import functional_hip as FH
    from transforms.gpu_pipeline import BatchedAugment, al_train_transforms
    from transforms.image_transform import RandomGaussianNoise
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    b, h, w = 5, 40, 64
    x = torch.rand(b, 1, h, w, generator=g).to(dev)
    lab = torch.randint(0, 3, (b, h, w), generator=g).to(dev)
    ap = torch.tensor([1, -1, 0, -1, 1], dtype=torch.int32, device=dev)
    sentinel = -7.0
    # blur / low-res / element-wise / noise / affine / elastic: skipped samples keep the sentinel, pass-through samples are copied
    out = torch.full_like(x, sentinel)
    sg = torch.full((b,), 0.8, device=dev)
    ks = torch.full((b,), 3, dtype=torch.int32, device=dev)
    call("mia_gaussian_blur", _p(x), _p(out), b, 1, h, w, _p(sg), _p(ks), 3, _p(ap), _stream())
    assert (out[1] == sentinel).all() and (out[3] == sentinel).all() and torch.equal(out[2], x[2]) and not torch.equal(out[0], x[0])
    out = torch.full_like(x, sentinel)
    lw = torch.tensor([[20, 32]] * b, dtype=torch.int32, device=dev)
    call("mia_resize_bilinear", _p(x), _p(out), b, 1, h, w, h, w, _p(lw), _p(ap), _stream())
    assert (out[1] == sentinel).all() and torch.equal(out[2], x[2]) and not torch.equal(out[4], x[4])
    out = torch.full_like(x, sentinel)
    p0 = torch.full((b,), 1.3, device=dev)
    call("mia_elementwise", _p(x), _p(out), _c_i64(h * w), b, 0, _p(p0), None, None, _p(ap), _stream())
    assert (out[3] == sentinel).all() and torch.equal(out[2], x[2]) and torch.allclose(out[0], x[0] ** 1.3, atol=1e-6)
    out = torch.full_like(x, sentinel)
    call("mia_noise_clip", _p(x), _p(out), _c_i64(h * w), b, _p(sg), ctypes.c_uint64(5), ctypes.c_uint64(0), _p(ap), _stream())
    assert (out[1] == sentinel).all() and (out[3] == sentinel).all() and torch.equal(out[2], x[2]) and not torch.equal(out[0], x[0])
    oi, ol = torch.full_like(x, sentinel), torch.full_like(lab, -7)
    mats = torch.tensor([[0.9, 0.1, 0.0, -0.1, 0.9, 0.0]] * b, device=dev)
    call("mia_affine_nearest", _p(x), _p(oi), _p(lab), _p(ol), b, 1, h, w, _p(mats), _p(ap), _stream())
    assert (oi[1] == sentinel).all() and (ol[3] == -7).all() and torch.equal(oi[2], x[2]) and torch.equal(ol[2], lab[2])
    oi, ol = torch.full_like(x, sentinel), torch.full_like(lab, -7)
    disp = torch.randn(b, 2, 3, 3, generator=g).to(dev) * 3
    call("mia_elastic_warp", _p(x), _p(oi), _p(lab), _p(ol), b, 1, h, w, _p(disp), 3, 3, _p(ap), _stream())
    assert (oi[3] == sentinel).all() and (ol[1] == -7).all() and torch.equal(oi[2], x[2]) and torch.equal(ol[2], lab[2])
    # copy_selected / sample_stats_sel
    dst = torch.full_like(x, sentinel)
    call("mia_copy_selected", _p(x), _p(dst), _c_i64(h * w * 4), b, _p(ap), _stream())
    assert torch.equal(dst[0], x[0]) and torch.equal(dst[4], x[4]) and (dst[1:4] == sentinel).all()
    ws = torch.empty(FH.lib().mia_sample_stats_workspace(b), device=dev)
    ms = torch.empty(b, 2, device=dev)
    call("mia_sample_stats_sel", _p(x), b, 1, _c_i64(h * w), 0, _p(ws), _p(ms), _p(ap), _stream())
    full = FH.sample_stats(x)
    assert torch.equal(ms[[0, 2, 4]], full[[0, 2, 4]]) and (ms[[1, 3]] == 0).all()
    # pipeline: selected-sample mode == whole-batch mode, inputs preserved, inplace identical
    RandomGaussianNoise.exact_rng = True
    try:
        imgs = torch.rand(8, 1, 96, 128, generator=g).to(dev)
        labs = torch.randint(0, 3, (8, 96, 128), generator=g).to(dev)
        keep_i, keep_l = imgs.clone(), labs.clone()
        pipe = al_train_transforms("busi", elastic=True)
        seed = next(s for s in range(300, 400) if _n_stages(pipe, s, 8, (1, 96, 128)) >= 6)
        torch.manual_seed(seed)
        got = BatchedAugment(pipe, image_size=64)(imgs, labs)
        assert torch.equal(imgs, keep_i) and torch.equal(labs, keep_l)
        # whole-batch reference: the same pipeline with the selective switch held off
        real = FH.set_selective
        FH.set_selective = lambda flag: real(False)
        try:
            torch.manual_seed(seed)
            want = BatchedAugment(pipe, image_size=64)(imgs, labs)
        finally:
            FH.set_selective = real
        assert torch.equal(got["image"], want["image"]) and torch.equal(got["label"], want["label"])
        assert 0 < got["_bytes"] < want["_bytes"] or got["_bytes"] == want["_bytes"]
        torch.manual_seed(seed)
        inpl = BatchedAugment(pipe, image_size=64, inplace=True)(imgs.clone(), labs.clone())
        assert torch.equal(inpl["image"], got["image"]) and torch.equal(inpl["label"], got["label"])
    finally:
        RandomGaussianNoise.exact_rng = False


def _n_stages(pipe, seed, b, shape):
    torch.manual_seed(seed)
    return sum(1 for _ in range(b) for q in pipe.draw(shape) if q is not None)
