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


def test_quick_gradient_on_the_table_and_random_cases(g):
    """quick_pairwise_gradient_zero_margin (counting kernel) on the reference's margin-0 table rows and on the
    random cases: gradient, max_reversals and total_pairs are exact integers"""
    from seesaw_amd import rank_loss as rl
    n_checked = 0
    for i in range(int(g["n_table"])):
        if float(g[f"t{i}_margin"]) != 0.0 or g[f"t{i}_target"].shape[0] == 0:
            continue
        grad, mx, _ = rl.quick_pairwise_gradient_zero_margin(g[f"t{i}_target"], scores=g[f"t{i}_scores"], return_max_inversions=True)
        assert np.array_equal(grad, 2.0 * np.asarray(g[f"t{i}_expected_gradient"], dtype=np.float32).reshape(-1)), i
        assert np.array_equal(mx, np.asarray(g[f"t{i}_expected_max_inversions"], dtype=np.float32).reshape(-1)), i
        n_checked += 1
    assert n_checked >= 4
    for i in range(int(g["n_random"])):
        grad, mx, total = rl.quick_pairwise_gradient_zero_margin(g[f"r{i}_target"], scores=g[f"r{i}_scores"], return_max_inversions=True)
        assert np.array_equal(grad, g[f"r{i}_quick_grad"]), i       # scores rounded to 1 decimal: many exact ties
        assert np.array_equal(mx, g[f"r{i}_quick_maxrev"]) and total == int(g[f"r{i}_quick_total"])
        loss, back = rl.cheap_pairwise_rank_loss(g[f"r{i}_target"], scores=g[f"r{i}_scores"])
        assert np.allclose(loss, g[f"r{i}_cheap_loss"], rtol=1e-6, atol=0)
    assert rl.quick_pairwise_gradient_zero_margin(np.zeros(0), scores=np.zeros(0)).shape == (0,)


def test_compute_inversions_matches_reference(g):
    from seesaw_amd.pairwise_rank_loss import compute_inversions
    for i in range(4):
        got = compute_inversions(g[f"inv{i}_labs"], g[f"inv{i}_scores"])
        assert np.array_equal(got, g[f"inv{i}_inversions"]), i


def test_rank_and_loss_and_vecstate_match_reference(g, oracle):
    from seesaw_amd.pairwise_rank_loss import VecState, rank_and_loss
    for k in range(int(g["n_ral"])):
        seed, n, n_pos = (int(v) for v in g[f"ral{k}_set"])
        X, _, q = oracle.labelled_set(seed, n, n_pos, q_noise=3.0)
        y = g[f"ral{k}_y"]
        loss, grad = rank_and_loss(q, X, y, float(g[f"ral{k}_margin"]))
        assert abs(loss - float(g[f"ral{k}_loss"])) <= 1e-6, (k, loss, float(g[f"ral{k}_loss"]))
        assert np.abs(grad - g[f"ral{k}_grad"]).max() <= 1e-6, k
    seed, n, n_pos = (int(v) for v in g["vs_set"])
    X, y, q = oracle.labelled_set(seed, n, n_pos)
    vs = VecState(q.copy(), margin=0.1, opt_params={"lr": 0.01}, renormalize=True)
    for step in range(3):
        vs.update(X, y)
        assert np.abs(vs.get_vec() - g[f"vs_w{step}"]).max() <= 1e-6, step


def test_rank_regression_lossgrad_along_reference_trajectory_and_fit(g, oracle):
    """RankRegressionPT: the loss (net inversions / total_pairs + regulariser) and the pseudo-gradient at every
    point the reference's L-BFGS evaluated, within 1e-4; and a fit from the reference's own start weights ends
    at a loss no worse than the reference's"""
    from seesaw_amd.logistic_regression import RankRegressionPT
    for k in range(int(g["n_rr"])):
        seed, n, n_pos = (int(v) for v in g[f"rr{k}_set"])
        X, y, q = oracle.labelled_set(seed, n, n_pos, q_noise=2.0)
        model = RankRegressionPT(scale="centered", reg_lambda=float(g[f"rr{k}_lam"]), regularizer_vector=q, max_iter=60, lr=1.0)
        model.fit(X, y.reshape(-1, 1), w0=g[f"rr{k}_w0"].reshape(-1))
        W, L, G = g[f"rr{k}_traj_w"], g[f"rr{k}_traj_loss"], g[f"rr{k}_traj_grad"]
        for t in range(W.shape[0]):
            loss, grad, _ = model.lossgrad(W[t])
            assert abs(loss - L[t]) <= TOL * max(1.0, abs(L[t])), (k, t, loss, L[t])
            assert np.abs(grad - G[t]).max() <= TOL * max(1.0, np.abs(G[t]).max()), (k, t)
        ours, _, _ = model.lossgrad(model.get_coeff().reshape(-1))
        assert ours <= L[-1] + 1e-4, (k, ours, L[-1])
        Xc = X - X.mean(axis=0)
        print(f"rank regression {k}: final loss ours {ours:.3e} vs reference {L[-1]:.3e}; |scores diff| = "
              f"{np.abs(Xc @ (model.get_coeff().reshape(-1) - g[f'rr{k}_coeff'].reshape(-1))).max():.2e}")


def test_rank_regression_narrow_dim_equals_zero_padded_wide_dim():
    """dim = 16 with 300 rows: the rank objective's [n] scratch is larger than the [n / 32, dim] partial-gradient
    buffer it once borrowed (out-of-bounds for dim < 32).  The same data zero-padded to dim = 32 is the
    same objective, so loss and gradient must agree and the padded gradient entries must vanish."""
    from seesaw_amd.logistic_regression import RankRegressionPT
    rng = np.random.default_rng(5)
    n, d = 300, 16
    X = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    q /= np.linalg.norm(q)
    y = (X @ q + 0.5 * rng.standard_normal(n) > 0.8).astype(np.float32)
    Xp = np.concatenate([X, np.zeros((n, 16), np.float32)], axis=1)
    qp = np.concatenate([q, np.zeros(16, np.float32)])
    narrow = RankRegressionPT(scale="centered", reg_lambda=1.0, regularizer_vector=q, max_iter=20, lr=1.0)
    wide = RankRegressionPT(scale="centered", reg_lambda=1.0, regularizer_vector=qp, max_iter=20, lr=1.0)
    narrow.fit(X, y.reshape(-1, 1), w0=q.copy())
    wide.fit(Xp, y.reshape(-1, 1), w0=qp.copy())
    for t in range(4):
        w = (q + 0.3 * rng.standard_normal(d)).astype(np.float32)
        l1, g1, _ = narrow.lossgrad(w)
        l2, g2, _ = wide.lossgrad(np.concatenate([w, np.zeros(16, np.float32)]))
        assert abs(l1 - l2) <= 1e-6 * max(1.0, abs(l2)), (t, l1, l2)
        assert np.abs(g1[:d] - g2[:d]).max() <= 1e-6 * max(1.0, np.abs(g2).max()), t
        assert np.abs(g2[d:32]).max() == 0.0
    assert np.abs(narrow.get_coeff().reshape(-1)[:d] - wide.get_coeff().reshape(-1)[:d]).max() <= 1e-5
