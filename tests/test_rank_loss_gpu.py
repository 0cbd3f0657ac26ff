"""GPU: the pairwise rank-loss kernel (k_fb_pairwise, the one the MultiReg fit evaluates) run directly on the
reference's own known-answer table (seesaw/test_rank_loss.py:9-234, 17 cases) and on seeded random cases,
against values the reference's functions returned (tests/golden/rank_loss.npz).
Tolerance: north_star's 1e-4 (f32); the integer-valued table entries come out exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "rank_loss.npz"))


def test_known_answer_table_through_the_fit_kernel(g):
    from seesaw_amd import rank_loss as rl
    n_table = int(g["n_table"])
    assert n_table == 17
    for i in range(n_table):
        target, scores, margin = g[f"t{i}_target"], g[f"t{i}_scores"], float(g[f"t{i}_margin"])
        n = target.shape[0]
        # what the table itself states (expected_*), and what the reference's functions returned on it
        exp_loss = np.asarray(g[f"t{i}_expected_rank_loss"], dtype=np.float64)
        exp_grad = 2.0 * np.asarray(g[f"t{i}_expected_gradient"], dtype=np.float64)
        exp_max = np.asarray(g[f"t{i}_expected_max_inversions"], dtype=np.float64).reshape(-1)
        ref_loss = np.asarray(g[f"t{i}_loss"], dtype=np.float64)
        ref_grad = np.asarray(g[f"t{i}_grad"], dtype=np.float64)
        col, mx = rl.ref_pairwise_rank_loss(target, scores=scores, margin=margin, aggregate="sum",
                                            return_max_inversions=True)
        grad = rl.ref_pairwise_rank_loss_gradient(target, scores=scores, margin=margin)
        assert col.shape == (n,) and grad.shape == (n,)
        if n == 0:
            continue
        assert np.array_equal(mx, exp_max), (i, mx, exp_max)
        ref_col = ref_loss.reshape(n, n).sum(0) if ref_loss.ndim == 2 or ref_loss.size == n * n else ref_loss
        assert np.abs(col - ref_col).max() <= TOL, (i, col, ref_col)
        assert np.abs(grad - ref_grad).max() <= TOL, (i, grad, ref_grad)
        assert np.abs(grad - exp_grad.reshape(-1)).max() <= TOL, (i, grad, exp_grad)
        if exp_loss.size == n * n:
            assert np.abs(col - exp_loss.reshape(n, n).sum(0)).max() <= TOL, i


def test_random_cases_hinge_and_logistic(g):
    from seesaw_amd import rank_loss as rl
    for i in range(int(g["n_random"])):
        target, scores, margin = g[f"r{i}_target"], g[f"r{i}_scores"], float(g[f"r{i}_margin"])
        col, mx = rl.ref_pairwise_rank_loss(target, scores=scores, margin=margin, return_max_inversions=True)
        assert np.array_equal(mx, g[f"r{i}_max_inv"].astype(np.float32)), i
        scale = max(1.0, np.abs(g[f"r{i}_hinge_sum"]).max())
        assert np.abs(col - g[f"r{i}_hinge_sum"]).max() <= TOL * scale, i
        grad = rl.ref_pairwise_rank_loss_gradient(target, scores=scores, margin=margin)
        assert np.abs(grad - g[f"r{i}_hinge_grad"]).max() <= TOL * max(1.0, np.abs(g[f"r{i}_hinge_grad"]).max()), i
        lcol = rl.ref_pairwise_logistic_loss(target, scores=scores)
        assert np.abs(lcol - g[f"r{i}_logistic_sum"]).max() <= TOL * max(1.0, np.abs(g[f"r{i}_logistic_sum"]).max()), i


def test_normalised_form_is_what_regmodule_sums(g):
    """coef = sample weights: item_j = sw_j * column_j / max_inversions_j (multi_reg.py:106-121)"""
    from seesaw_amd import rank_loss as rl
    i = 1
    target, scores, margin = g[f"r{i}_target"], g[f"r{i}_scores"], float(g[f"r{i}_margin"])
    sw = np.linspace(0.5, 2.0, target.shape[0]).astype(np.float32)
    item, _ = rl.pairwise_sums(target, scores=scores, margin=margin, coef=sw)
    mx = g[f"r{i}_max_inv"].astype(np.float64)
    want = np.where(mx > 0, sw * g[f"r{i}_hinge_sum"] / np.maximum(mx, 1), 0.0)
    assert np.abs(item - want).max() <= TOL
