"""CPU restatement of the active-learning selectors' arithmetic (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

* acquisition scores: entropy_selector.py:42-49, confidence_selector.py:42-47, margin_selector.py:42-48
* k-centre greedy: coreset_selector.py:19-52, restated literally (re-slices the distance matrix every round)
* row standardisation of encoder features: kmean_selector.py:98-104
PINNED (round 4): `oracle/_refload.Ref.load_selectors` imports the reference's own selector modules (empty placeholders for
`h5py` -- only the `feature_path` option uses it -- and for the `datasets.active_dataset` type annotation), and
`oracle/gen_golden.gen_selectors` drives them with the reference UNet on a synthetic pool -> `tests/golden/selectors.npz`;
`tests/test_oracle_golden.py::test_selectors_oracle_vs_reference_vectors` checks these restatements against it.  Still
unpinned: features read from `feature_path` *.h5 files (h5py is not importable here).
"""
from __future__ import annotations

import numpy as np
import torch


def entropy_score(logits: torch.Tensor, smooth: float = 1e-8) -> torch.Tensor:
    prob = logits.softmax(1)
    return torch.mean(-prob * torch.log2(prob + smooth), dim=1).mean(dim=[-2, -1])


def confidence_score(logits: torch.Tensor) -> torch.Tensor:
    return (-1 * logits.softmax(1).max(1)[0]).mean(dim=[-2, -1])


def margin_score(logits: torch.Tensor) -> torch.Tensor:
    top2 = torch.topk(logits.softmax(1), k=2, dim=1)[0]
    return (-1 * (top2[:, 0] - top2[:, 1])).mean(dim=[-2, -1])


def kcenter_greedy(dist_mat: np.ndarray, n_data: int, budget: int, init_idx, coreset_criteria: str = "min") -> list:
    all_indices = np.arange(n_data)
    labeled = np.zeros((n_data,), dtype=np.bool_)
    labeled[init_idx] = True
    for _ in range(budget):
        mat = dist_mat[~labeled, :][:, labeled]
        red = mat.min(axis=1) if coreset_criteria == "min" else mat.mean(axis=1)
        labeled[all_indices[~labeled][red.argmax()]] = True
    return list(set(all_indices[labeled]) - set(init_idx))


def row_standardise(f: np.ndarray) -> np.ndarray:
    return (f - np.mean(f, axis=1, keepdims=True)) / np.std(f, axis=1, keepdims=True)
