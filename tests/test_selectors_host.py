"""Host logic of the active-learning selectors (CPU): incremental k-centre greedy vs the literal restatement of
coreset_selector.py:19-52, random pick, feature standardisation."""
import numpy as np
import torch

from activelearning.selectors import RandomSelector, _row_standardise, kcenter_greedy
from oracle import selectors_ref


def test_kcenter_greedy_matches_literal_restatement():
    rng = np.random.default_rng(0)
    for n, m, budget in ((30, 4, 6), (50, 1, 10), (17, 8, 9)):
        x = rng.normal(size=(n, 5))
        d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
        init = np.arange(m)
        for crit in ("min", "mean"):
            got = kcenter_greedy(d, n, budget, init, crit)
            want = selectors_ref.kcenter_greedy(d, n, budget, init, crit)
            assert sorted(got) == sorted(int(v) for v in want), (n, m, budget, crit)
            assert len(got) == budget and not set(got) & set(init.tolist())


def test_kcenter_greedy_ties_take_lowest_index():
    d = np.ones((6, 6)) - np.eye(6)  # every unlabelled point is equally far
    assert sorted(kcenter_greedy(d, 6, 2, np.array([3]), "min")) == sorted(selectors_ref.kcenter_greedy(d, 6, 2, np.array([3]), "min"))


def test_random_selector_and_standardise():
    class _Pool:
        image_idx = [f"case_{i}" for i in range(9)]

    class _AD:
        pool_dataset = _Pool()

        def get_size(self):
            return 0, 9

    torch.manual_seed(3)
    got = RandomSelector().select_next_batch(_AD(), 4, None, None)
    torch.manual_seed(3)
    _, idx = torch.sort(torch.rand(9), descending=True)
    assert got == [f"case_{int(i)}" for i in idx[:4]]
    f = np.random.default_rng(1).normal(size=(5, 12)) * 3 + 2
    np.testing.assert_allclose(_row_standardise(f), selectors_ref.row_standardise(f))
    np.testing.assert_allclose(_row_standardise(f).std(axis=1), 1.0, rtol=1e-12)
