"""Active-learning selectors on the MI355X forward path (reference `src/activelearning/*.py`; constructed by
`ALTrainer._setup_active_selector`, al_trainer.py:802-883, called once per round at :1061-1066).

Same class names, constructor arguments and ``select_next_batch(active_dataset, select_num, model, device)`` contract.
The per-image arithmetic runs in HIP kernels (``selector_scores``: fused softmax + entropy / confidence / margin
reduction; ``model.get_enc_feature``: NHWC global average pool); clustering / distance matrices stay on the host with
scikit-learn exactly as in the reference (kmean_selector.py:143-145,190-192, coreset_selector.py:122-167).

``active_dataset`` is duck-typed like the reference uses it: ``get_size() -> (labeled, pool)``, ``get_pool_dataset()``,
``get_train_dataset()``, ``pool_dataset.image_idx``; datasets yield dicts with ``"image"`` and ``"case_name"``.
Features loaded from ``feature_path`` (*.h5, kmean_selector.py:77-86) need h5py, which is imported on first use.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Callable, Optional

import numpy as np
import torch
from torch.utils.data import ConcatDataset, DataLoader

from .scores import CONFIDENCE, ENTROPY, MARGIN, selector_scores


class ActiveSelector(ABC):
    """active_selector.py:10-19"""

    @abstractmethod
    def select_next_batch(self, active_dataset, select_num: int, model, device) -> list:
        ...


def _random_pick(active_dataset, select_num: int) -> list:
    """torch.rand over the pool, sort descending, first `select_num` (random_selector.py:17-23)."""
    _, pool_size = active_dataset.get_size()
    scores = torch.rand(pool_size)
    _, indices = torch.sort(scores, descending=True)
    return [active_dataset.pool_dataset.image_idx[i] for i in indices[:select_num]]


class RandomSelector(ActiveSelector):
    def select_next_batch(self, active_dataset, select_num, model=None, device=None):
        return _random_pick(active_dataset, select_num)


def _loader(dataset, batch_size, num_workers, pin_memory):
    return DataLoader(dataset=dataset, batch_size=batch_size, num_workers=num_workers, pin_memory=pin_memory)


class _ScoreSelector(ActiveSelector):
    """Shared body of the three uncertainty selectors: score every pool image, take the `select_num` largest."""
    score_column = ENTROPY

    def __init__(self, batch_size: int, num_workers: int, pin_memory: bool = True, smooth: float = 1e-8) -> None:
        self.batch_size, self.num_workers, self.pin_memory, self.smooth = batch_size, num_workers, pin_memory, smooth

    def cal_scores(self, active_dataset, model, device):
        score_list, case_name_list = [], []
        model.eval()
        for batch in _loader(active_dataset.get_pool_dataset(), self.batch_size, self.num_workers, self.pin_memory):
            with torch.no_grad():
                scores = selector_scores(model(batch["image"].to(device)), self.smooth)[:, self.score_column]
            score_list.extend(scores)
            case_name_list.extend(batch["case_name"])
        return score_list, case_name_list

    def select_next_batch(self, active_dataset, select_num, model, device):
        labeled_size, _ = active_dataset.get_size()
        if labeled_size == 0:
            return _random_pick(active_dataset, select_num)
        score_list, case_name_list = self.cal_scores(active_dataset, model, device)
        _, indices = torch.sort(torch.stack(score_list, dim=0), descending=True)
        return [case_name_list[i] for i in indices[:select_num]]


class EntropySelector(_ScoreSelector):
    """mean_{c,h,w}(-p log2(p + smooth)) (entropy_selector.py:42-49)"""
    score_column = ENTROPY


class ConfidenceSelector(_ScoreSelector):
    """mean_{h,w}(-max_c p) (confidence_selector.py:42-47)"""
    score_column = CONFIDENCE


class MarginSelector(_ScoreSelector):
    """mean_{h,w}(-(p_top1 - p_top2)) (margin_selector.py:42-48)"""
    score_column = MARGIN


def _row_standardise(f: np.ndarray) -> np.ndarray:
    return (f - np.mean(f, axis=1, keepdims=True)) / np.std(f, axis=1, keepdims=True)


def _read_h5_feature(feature_path: Path, case: str) -> np.ndarray:
    import h5py  # not in every image; only the feature_path option needs it
    with h5py.File(feature_path / f"{case}.h5", "r") as h5f:
        return h5f["feature"][:]


def _encoder_and_loaded_features(dataset, model, device, batch_size, num_workers, pin_memory, feature_path, feature_dict,
                                 want_model: bool):
    feats, loaded, names = [], [], []
    for batch in _loader(dataset, batch_size, num_workers, pin_memory):
        case_name = batch["case_name"]
        names.extend(case_name)
        if want_model:
            model.eval()
            with torch.no_grad():
                feats.append(model.get_enc_feature(batch["image"].to(device)).cpu().numpy())
        if feature_path:
            loaded.extend(_read_h5_feature(feature_path, c) for c in case_name)
        elif feature_dict:
            loaded.extend(feature_dict[c] for c in case_name)
    return (np.concatenate(feats, axis=0) if feats else None), (np.stack(loaded, axis=0) if loaded else None), names


class KMeanSelector(ActiveSelector):
    """k-means++ seeding over row-standardised encoder features, weighted by the distance to the labelled set
    (kmean_selector.py:20-196)."""

    def __init__(self, batch_size: int, num_workers: int, pin_memory: bool = True, smooth: float = 1e-8, metric: str = "cosine",
                 feature_path=None, feature_dict: Optional[dict] = None, coreset_criteria: str = "min",
                 loaded_feature_weight: float = 1.0, loaded_feature_only: bool = False, sharp_factor: float = 1.0,
                 softmax: bool = False) -> None:
        self.batch_size, self.num_workers, self.pin_memory, self.smooth = batch_size, num_workers, pin_memory, smooth
        self.metric = metric
        self.feature_path = Path(feature_path) if feature_path else None
        self.feature_dict = feature_dict
        self.coreset_criteria = coreset_criteria
        self.loaded_feature_weight, self.loaded_feature_only = loaded_feature_weight, loaded_feature_only
        self.sharp_factor, self.softmax = sharp_factor, softmax

    def get_features(self, dataset, model, device):
        want_model = bool(model) and not self.loaded_feature_only
        feats, loaded, names = _encoder_and_loaded_features(dataset, model, device, self.batch_size, self.num_workers,
                                                            self.pin_memory, self.feature_path, self.feature_dict, want_model)
        parts = []
        if feats is not None:
            feats = _row_standardise(feats)
            parts.append(feats)
        if loaded is not None:
            loaded = _row_standardise(loaded)
            scale = 1 if feats is None else np.sqrt(feats.shape[-1] / loaded.shape[-1] * self.loaded_feature_weight)
            parts.append(loaded * scale)
        return np.concatenate(parts, axis=1), np.array(names)

    def cal_scores(self, active_dataset, model, device):
        from sklearn.metrics import pairwise_distances
        labeled_size, _ = active_dataset.get_size()
        pool_feats, pool_names = self.get_features(active_dataset.get_pool_dataset(), model, device)
        if labeled_size > 0:
            labeled_feats, labeled_names = self.get_features(active_dataset.get_train_dataset(), model, device)
            dist = pairwise_distances(pool_feats, labeled_feats, metric=self.metric)
        else:
            labeled_feats = labeled_names = dist = None
        return labeled_feats, pool_feats, labeled_names, pool_names, dist

    def select_next_batch(self, active_dataset, select_num, model, device):
        from sklearn.cluster import kmeans_plusplus
        _, pool_feats, _, pool_names, dist = self.cal_scores(active_dataset, model, device)
        weight = None
        if dist is not None:
            weight = dist.min(axis=1) if self.coreset_criteria == "min" else dist.mean(axis=1)
            if self.softmax:
                weight = (torch.from_numpy(weight) * self.sharp_factor).softmax(0).numpy()
            else:
                weight = weight ** self.sharp_factor
                weight = weight / weight.sum()
        _, picked = kmeans_plusplus(X=pool_feats, n_clusters=select_num, sample_weight=weight)
        return list(set(pool_names[picked].tolist()))


def kcenter_greedy(dist_mat: np.ndarray, n_data: int, budget: int, init_idx, coreset_criteria: str = "min") -> list:
    """Greedy k-centre (coreset_selector.py:19-52): `budget` times, add the unlabelled point whose distance to the
    labelled set (min, or mean, over labelled columns) is largest; ties -> lowest index, like ``argmax`` over the
    index-ordered unlabelled rows.  The reference re-slices the matrix every round (O(budget * n * m)); this keeps the
    running min (or sum) per row and folds in one column per pick -- identical picks for "min" (min is exact); for
    "mean" the running sum can differ from numpy's pairwise mean in the last ulp."""
    assert dist_mat.shape[0] == n_data, "Size of distance matrix and number of data doesn't match!"
    if coreset_criteria not in ("min", "mean"):
        raise RuntimeError(f"coreset_criteria {coreset_criteria} is undefined")
    labeled = np.zeros((n_data,), dtype=np.bool_)
    labeled[init_idx] = True
    init_set = set(np.arange(n_data)[labeled].tolist())
    cols = dist_mat[:, labeled]
    count = int(labeled.sum())
    if coreset_criteria == "min":
        cur = cols.min(axis=1) if count else np.full((n_data,), np.inf)
    else:
        cur = cols.sum(axis=1, dtype=np.float64)
    for _ in range(budget):
        crit = cur if coreset_criteria == "min" else cur / max(count, 1)
        q = int(np.argmax(np.where(labeled, -np.inf, crit)))
        labeled[q] = True
        count += 1
        cur = np.minimum(cur, dist_mat[:, q]) if coreset_criteria == "min" else cur + dist_mat[:, q]
    return list(set(np.arange(n_data)[labeled].tolist()) - init_set)


class CoresetSelector(ActiveSelector):
    """k-centre greedy over encoder-feature distances (coreset_selector.py:55-232)."""

    def __init__(self, batch_size: int, num_workers: int, pin_memory: bool = True, smooth: float = 1e-8, metric: str = "cosine",
                 coreset_criteria: str = "min", coreset_fusion: str = "add", feature_path=None,
                 loaded_feature_weight: float = 0.0) -> None:
        self.batch_size, self.num_workers, self.pin_memory, self.smooth = batch_size, num_workers, pin_memory, smooth
        self.metric = metric
        self.feature_path = Path(feature_path) if feature_path else None
        self.coreset_criteria, self.coreset_fusion = coreset_criteria, coreset_fusion
        self.loaded_feature_weight = loaded_feature_weight

    def cal_scores(self, active_dataset, model, device):
        from sklearn.metrics import pairwise_distances
        labeled_ds, pool_ds = active_dataset.get_train_dataset(), active_dataset.get_pool_dataset()
        core_list = labeled_ds.image_idx
        all_list = labeled_ds.image_idx + pool_ds.image_idx
        feats, loaded, _ = _encoder_and_loaded_features(ConcatDataset([labeled_ds, pool_ds]), model, device, self.batch_size,
                                                        self.num_workers, self.pin_memory, self.feature_path, None, bool(model))
        if self.coreset_fusion == "add":
            final = 0
            if loaded is not None:
                d = pairwise_distances(loaded, metric=self.metric)
                final = final + self.loaded_feature_weight * (d / d.sum())
            if feats is not None:
                d = pairwise_distances(feats, metric=self.metric)
                final = final + (1 - self.loaded_feature_weight) * (d / d.sum())
        else:
            parts = [] if feats is None else [feats]
            if loaded is not None:
                scale = 1 if feats is None else np.sqrt(feats.shape[-1] / loaded.shape[-1] * self.loaded_feature_weight)
                parts.append(loaded * scale)
            final = pairwise_distances(np.concatenate(parts, axis=1), metric=self.metric)
        return np.array(core_list), np.array(all_list), loaded, feats, final

    def select_next_batch(self, active_dataset, select_num, model, device):
        from sklearn.cluster import kmeans_plusplus
        labeled_size, _ = active_dataset.get_size()
        if labeled_size == 0:
            if self.loaded_feature_weight == 0 or not self.feature_path:
                return _random_pick(active_dataset, select_num)
            _, all_list, loaded, _, _ = self.cal_scores(active_dataset, None, device)
            _, picked = kmeans_plusplus(X=loaded, n_clusters=select_num)
            return list(all_list[picked])
        core_list, all_list, _, _, dist = self.cal_scores(active_dataset, model, device)
        ids = kcenter_greedy(dist_mat=dist, n_data=len(all_list), budget=select_num, init_idx=np.arange(len(core_list)),
                             coreset_criteria=self.coreset_criteria)
        return list(all_list[ids])


def image_wise_grad(loss: torch.Tensor, model, last_layer_name: str = "decoder.seg_output.weight") -> torch.Tensor:
    """Gradient embedding = d loss / d (last layer weight), flattened (badge_selector.py:19-35).  The reference runs a full
    ``loss.backward(retain_graph=True)`` and reads one ``.grad``; asking autograd for that single tensor gives the same
    values and stops at the 1x1 head (``mia_head_bwd``) instead of walking the whole network."""
    model.zero_grad()
    last = dict(model.named_parameters())[last_layer_name]
    (g,) = torch.autograd.grad(loss, last, retain_graph=True)
    model.zero_grad()
    return g.detach().flatten().clone()


class BADGESelector(ActiveSelector):
    """k-means++ over last-layer gradient embeddings of the pseudo-labelled loss (badge_selector.py:38-128)."""

    def __init__(self, dice_loss: Callable, ce_loss: Callable, batch_size: int, num_workers: int, pin_memory: bool = True,
                 smooth: float = 1e-8, multiple_loss: str = "add") -> None:
        self.dice_loss, self.ce_loss = dice_loss, ce_loss
        self.batch_size, self.num_workers, self.pin_memory = batch_size, num_workers, pin_memory
        self.multiple_loss, self.smooth = multiple_loss, smooth

    def cal_scores(self, active_dataset, model, device):
        model.eval()
        embeds, names = [], []
        for batch in _loader(active_dataset.get_pool_dataset(), self.batch_size, self.num_workers, self.pin_memory):
            names.extend(batch["case_name"])
            outputs = model(batch["image"].to(device))
            preds = outputs.softmax(1).argmax(1)
            if self.multiple_loss == "sep":
                # the reference concatenates two 0-dim losses here (badge_selector.py:85-91), which torch rejects
                raise RuntimeError("zero-dimensional tensor (at position 0) cannot be concatenated")
            if isinstance(self.ce_loss, torch.nn.CrossEntropyLoss):  # al_train's choice: keep it on the fused HIP loss kernel
                from losses.ce_loss import hip_cross_entropy
                ce = hip_cross_entropy(self.ce_loss, outputs, preds)
            else:
                ce = self.ce_loss(outputs, preds)
            loss = ce + self.dice_loss(outputs, preds)
            embeds.append(image_wise_grad(loss, model))
        return np.array(names), torch.stack(embeds, dim=0).cpu().numpy()

    def select_next_batch(self, active_dataset, select_num, model, device):
        from sklearn.cluster import kmeans_plusplus
        labeled_size, _ = active_dataset.get_size()
        if labeled_size == 0:
            return _random_pick(active_dataset, select_num)
        names, embeds = self.cal_scores(active_dataset, model, device)
        _, picked = kmeans_plusplus(X=embeds, n_clusters=select_num)
        return list(names[picked])
