"""Active-learning selectors end to end on the HIP forward path: a tiny UNet scores a synthetic pool;
the picks must equal what the oracle's restatement of the reference formulas picks from the CPU oracle's logits."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _DS(torch.utils.data.Dataset):
    def __init__(self, images, names):
        self.images, self.image_idx = images, list(names)

    def __len__(self):
        return len(self.image_idx)

    def __getitem__(self, i):
        return {"image": self.images[i], "case_name": self.image_idx[i]}


class _ActiveDataset:
    """The four members the reference selectors touch (datasets/active_dataset.py)."""

    def __init__(self, images, n_labeled):
        names = [f"case_{i:02d}" for i in range(len(images))]
        self.train_dataset = _DS(images[:n_labeled], names[:n_labeled])
        self.pool_dataset = _DS(images[n_labeled:], names[n_labeled:])

    def get_size(self):
        return len(self.train_dataset), len(self.pool_dataset)

    def get_pool_dataset(self):
        return self.pool_dataset

    def get_train_dataset(self):
        return self.train_dataset


def _setup():
    from models.unet import UNet
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    model = UNet(2, 1, 3, [8, 16, 32], normalization="instance", dropout_prob=None).to(dev)
    g = torch.Generator().manual_seed(9)
    images = torch.rand(14, 1, 32, 32, generator=g)
    images *= torch.linspace(0.2, 3.0, 14).view(-1, 1, 1, 1)  # spread the uncertainty so the ranking is not a near-tie
    ad = _ActiveDataset(images, n_labeled=4)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    return dev, model, ad, images, params


def _oracle_logits(params, images):
    from oracle import unet_ref
    return unet_ref.unet_forward(params, images, normalization="instance", training=False)


def test_score_selectors_pick_what_the_oracle_picks():
    from activelearning import ConfidenceSelector, EntropySelector, MarginSelector
    from oracle import selectors_ref
    dev, model, ad, images, params = _setup()
    pool = images[4:]
    logits = _oracle_logits(params, pool)
    want = {"entropy": selectors_ref.entropy_score(logits), "confidence": selectors_ref.confidence_score(logits),
            "margin": selectors_ref.margin_score(logits)}
    for name, cls in (("entropy", EntropySelector), ("confidence", ConfidenceSelector), ("margin", MarginSelector)):
        sel = cls(batch_size=4, num_workers=0, pin_memory=False)
        scores, names = sel.cal_scores(ad, model, dev)
        got = torch.stack(scores).cpu()
        np.testing.assert_allclose(got.numpy(), want[name].numpy(), rtol=2e-4, atol=2e-6, err_msg=name)
        picks = sel.select_next_batch(ad, 3, model, dev)
        order = torch.sort(want[name], descending=True)[1][:3]
        assert picks == [ad.pool_dataset.image_idx[int(i)] for i in order], name
    # empty labelled set -> random pick from the pool (entropy_selector.py:62-70)
    ad0 = _ActiveDataset(images, n_labeled=0)
    torch.manual_seed(5)
    picks = EntropySelector(4, 0, False).select_next_batch(ad0, 5, model, dev)
    torch.manual_seed(5)
    idx = torch.sort(torch.rand(14), descending=True)[1][:5]
    assert picks == [ad0.pool_dataset.image_idx[int(i)] for i in idx]


def test_feature_selectors_and_badge():
    from activelearning import BADGESelector, CoresetSelector, KMeanSelector
    from losses.compound_losses import DiceAndCELoss
    from oracle import selectors_ref, unet_ref
    from sklearn.metrics import pairwise_distances
    dev, model, ad, images, params = _setup()
    # encoder features: HIP global-average-pool vs oracle
    feats_ref = unet_ref.enc_feature(params, images, normalization="instance", training=False).numpy()
    km = KMeanSelector(batch_size=5, num_workers=0, pin_memory=False, metric="l2")
    pool_feats, names = km.get_features(ad.get_pool_dataset(), model, dev)
    assert pool_feats.shape == (10, 32) and list(names) == ad.pool_dataset.image_idx
    np.testing.assert_allclose(pool_feats, selectors_ref.row_standardise(feats_ref[4:]), rtol=1e-3, atol=1e-4)
    picks = km.select_next_batch(ad, 3, model, dev)
    assert 1 <= len(picks) <= 3 and set(picks) <= set(ad.pool_dataset.image_idx)
    # coreset: same picks as the literal k-centre on the distance matrix of the HIP features
    cs = CoresetSelector(batch_size=5, num_workers=0, pin_memory=False, metric="l2")
    core, all_list, _, feats, dist = cs.cal_scores(ad, model, dev)
    assert list(all_list) == ad.train_dataset.image_idx + ad.pool_dataset.image_idx and feats.shape == (14, 32)
    d = pairwise_distances(feats, metric="l2")
    np.testing.assert_allclose(dist, d / d.sum(), rtol=1e-12)
    picks = cs.select_next_batch(ad, 4, model, dev)
    want = selectors_ref.kcenter_greedy(dist, 14, 4, np.arange(4), "min")
    assert sorted(picks) == sorted(all_list[want].tolist())
    # BADGE: gradient embedding of the pseudo-labelled Dice+CE loss w.r.t. decoder.seg_output.weight
    loss = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    bd = BADGESelector(dice_loss=loss.dice_loss, ce_loss=loss.ce_loss, batch_size=1, num_workers=0, pin_memory=False)
    bnames, embeds = bd.cal_scores(ad, model, dev)
    assert embeds.shape == (10, 3 * 8) and list(bnames) == ad.pool_dataset.image_idx
    # oracle: same quantity with torch autograd on the CPU restatement
    from oracle import losses_ref
    p = {k: v.clone().requires_grad_(k == "decoder.seg_output.weight") for k, v in params.items()}
    for i in range(3):
        out = unet_ref.unet_forward(p, images[4 + i:5 + i], normalization="instance", training=False)
        pred = out.softmax(1).argmax(1)
        l = losses_ref.ce_loss(out, pred) + losses_ref.dice_loss(out, pred, 2, do_bg=True)
        (gr,) = torch.autograd.grad(l, p["decoder.seg_output.weight"])
        np.testing.assert_allclose(embeds[i], gr.flatten().numpy(), rtol=2e-3, atol=2e-6)
    picks = bd.select_next_batch(ad, 3, model, dev)
    assert len(picks) == 3 and set(picks) <= set(ad.pool_dataset.image_idx)
    assert all(q.grad is None or float(q.grad.abs().sum()) == 0.0 for q in model.parameters())  # model.zero_grad() after each embed


def test_selectors_match_reference_vectors():
    """SURVEY 8(f)2, pinned: the HIP selectors against `tests/golden/selectors.npz`, written by the REFERENCE's own
    selector classes driving the reference UNet (`oracle/gen_golden.gen_selectors`): scores, picks, encoder features,
    distance matrices, k-centre picks, standardised k-means features, BADGE gradient embeddings."""
    from activelearning import (BADGESelector, ConfidenceSelector, CoresetSelector, EntropySelector, KMeanSelector,
                                MarginSelector)
    from activelearning.selectors import kcenter_greedy
    from losses.compound_losses import DiceAndCELoss
    from models.unet import UNet
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "selectors.npz"), allow_pickle=False))
    dev = torch.device("cuda:0")
    model = UNet(2, 1, 3, [8, 16, 32], normalization="instance", dropout_prob=None)
    model.load_state_dict({k[5:]: torch.from_numpy(v.copy()) for k, v in d.items() if k.startswith("init/")})
    model = model.to(dev)
    images = torch.from_numpy(d["images"])
    nl = int(d["n_labeled"])
    ad = _ActiveDataset(images, nl)
    for short, cls in (("entropy", EntropySelector), ("confidence", ConfidenceSelector), ("margin", MarginSelector)):
        sel = cls(batch_size=4, num_workers=0, pin_memory=False)
        scores, names = sel.cal_scores(ad, model, dev)
        np.testing.assert_allclose(torch.stack(scores).cpu().numpy(), d[f"{short}/scores"], rtol=2e-4, atol=2e-6, err_msg=short)
        assert list(names) == list(d[f"{short}/names"])
        assert sel.select_next_batch(ad, 3, model, dev) == list(d[f"{short}/picks3"]), short
    for metric, crit in (("cosine", "min"), ("l2", "min"), ("l2", "mean")):
        key = f"coreset_{metric}_{crit}"
        cs = CoresetSelector(batch_size=5, num_workers=0, pin_memory=False, metric=metric, coreset_criteria=crit)
        core, all_list, _, feats, dist = cs.cal_scores(ad, model, dev)
        assert list(core) == list(d[key + "/core"]) and list(all_list) == list(d[key + "/all"])
        np.testing.assert_allclose(feats, d[key + "/feats"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dist, d[key + "/dist"], rtol=2e-3, atol=1e-6)
        assert sorted(cs.select_next_batch(ad, 4, model, dev)) == sorted(d[key + "/picks4"].tolist()), key
    for crit in ("min", "mean"):  # the incremental k-centre against the reference's re-slicing one
        got = kcenter_greedy(d["kcenter/dist"], 24, 6, d["kcenter/init"].tolist(), crit)
        assert sorted(int(i) for i in got) == d[f"kcenter/{crit}_b6"].tolist(), crit
    km = KMeanSelector(batch_size=5, num_workers=0, pin_memory=False, metric="l2")
    lf, pf, ln, pn, p2l = km.cal_scores(ad, model, dev)
    np.testing.assert_allclose(pf, d["kmean/pool_feats"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(lf, d["kmean/labeled_feats"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(p2l, d["kmean/pool2labeled"], rtol=1e-3, atol=1e-3)
    assert list(pn) == list(d["kmean/pool_names"])
    np.random.seed(0)  # kmeans_plusplus(random_state=None) draws from numpy's global generator, as in the reference run
    assert sorted(km.select_next_batch(ad, 3, model, dev)) == d["kmean/picks3_npseed0"].tolist()
    loss = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    bd = BADGESelector(dice_loss=loss.dice_loss, ce_loss=loss.ce_loss, batch_size=1, num_workers=0, pin_memory=False)
    bn, be = bd.cal_scores(ad, model, dev)
    assert list(bn) == list(d["badge/names"])
    np.testing.assert_allclose(be, d["badge/embeds"], rtol=2e-3, atol=2e-6)
    ad0 = _ActiveDataset(images, 0)
    torch.manual_seed(5)
    assert EntropySelector(4, 0, False).select_next_batch(ad0, 5, model, dev) == list(d["entropy/picks5_empty_seed5"])
