"""CPU: the L-KNN model's host logic (seesaw/loops/LKNN_model.py) against the reference's ring-graph answers
(loops/LKNN_model_test.py:19-45) and the numpy oracle of the two-step look-ahead (_top_sum) against values the
reference returned over a planning session (tests/golden/lknn.npz) -- the session state is rebuilt with OUR model,
so its conditioning bookkeeping is pinned too."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _ring_model():
    from seesaw_amd.loops.LKNN_model import LKNNModel
    from seesaw_amd.research.active_search.common import Dataset
    mat = np.zeros((5, 5))
    for i in range(5):
        mat[i, (i + 1) % 5] = 1
    return LKNNModel.from_dataset(Dataset.from_vectors(np.zeros((5, 10))), weight_matrix=sp.csr_array(mat + mat.T),
                                  gamma=np.full(5, 0.5))


def test_ring_graph_known_answers():
    g = np.load(os.path.join(GOLDEN, "lknn.npz"))
    model = _ring_model()
    pts = np.arange(5)
    assert np.isclose(model.predict_proba(pts), 0.5).all()
    up = model.condition(2, 1)
    ids, probs = up.top_k_remaining(top_k=4)
    assert np.array_equal(ids, g["ring_cond1_ids"]) and np.array_equal(probs, g["ring_cond1"])
    # the table of the reference's test: [.5, .75, 1., .75, .5] with node 2 labelled (it is no longer "remaining")
    assert dict(zip(ids.tolist(), probs.tolist())) == {1: 0.75, 3: 0.75, 0: 0.5, 4: 0.5}
    assert np.isclose(model.predict_proba(pts), 0.5).all()          # no mutation
    ids0, probs0 = model.condition(2, 0).top_k_remaining(top_k=4)
    assert np.array_equal(ids0, g["ring_cond0_ids"]) and np.array_equal(probs0, g["ring_cond0"])
    assert 0.75 <= model.probability_bound(1) and 2.5 / 3 <= model.probability_bound(2)
    assert np.array_equal(np.array([model.probability_bound(1), model.probability_bound(2)]), g["ring_bounds"])
    model.condition_(2, 1)                                           # in place
    assert np.array_equal(model.score, np.array([0.5, 0.75, 0.5, 0.75, 0.5]))
    assert 2 in model.dataset.seen_indices and 2 not in model.dataset.remaining_indices()


def session_graph(g):
    N, D = int(g["N"]), int(g["D"])
    rng = np.random.default_rng(int(g["graph_seed"]))
    nbr = np.stack([rng.choice(N, D, replace=False) for _ in range(N)]).astype(np.int32)
    W = sp.csr_array((np.ones(N * D), nbr.reshape(-1), np.arange(0, N * D + 1, D)), shape=(N, N))
    truth = (rng.random(N) < 0.08).astype(np.int64)
    assert np.array_equal(truth, g["truth"])
    return N, D, nbr, W, truth


@pytest.mark.parametrize("horizon", [2, 9, 20, 101])
def test_oracle_top_sum_bit_exact_over_the_reference_session(oracle, horizon):
    from seesaw_amd.loops.LKNN_model import LKNNModel, initial_gamma_array
    from seesaw_amd.research.active_search.common import Dataset
    g = np.load(os.path.join(GOLDEN, "lknn.npz"))
    N, D, nbr, W, truth = session_graph(g)
    model = LKNNModel.from_dataset(Dataset.from_vectors(np.zeros((N, 1))), weight_matrix=W, gamma=initial_gamma_array(0.1, N))
    picks = g[f"h{horizon}_picks"]
    for rnd in range(12):
        if rnd in (0, 5, 11):
            numer = model.numerators + model.gamma
            numer[np.asarray(model.dataset.seen_indices, dtype=np.int64)] = -np.inf
            vals = oracle.lknn_top_sum(numer, model.denominators + 1, np.sort(nbr), horizon - 1)
            ref = g[f"h{horizon}_values_r{rnd}"]
            assert np.array_equal(np.isnan(vals), np.isnan(ref))
            ok = ~np.isnan(ref)
            assert np.array_equal(vals[ok].view(np.uint64), ref[ok].view(np.uint64)), (horizon, rnd)
            assert int(np.nanargmax(vals)) == int(picks[rnd]) and np.nanmax(vals) == g[f"h{horizon}_values"][rnd]
        model.condition_(int(picks[rnd]), int(truth[int(picks[rnd])]))


@pytest.mark.parametrize("horizon", [9, 101])
def test_host_fallback_of_the_look_ahead_equals_reference_values(horizon):
    """LKNNModel._top_sum_host -- what top_sum runs for horizons / degrees beyond the kernel's registers (K > 128, more
    than 32 neighbours: ADVICE r2) -- against the reference's value vectors of the planning session, bit for bit"""
    from seesaw_amd.loops.LKNN_model import LKNNModel, initial_gamma_array
    from seesaw_amd.research.active_search.common import Dataset
    g = np.load(os.path.join(GOLDEN, "lknn.npz"))
    N, D, nbr, W, truth = session_graph(g)
    model = LKNNModel.from_dataset(Dataset.from_vectors(np.zeros((N, 1))), weight_matrix=W, gamma=initial_gamma_array(0.1, N))
    picks = g[f"h{horizon}_picks"]
    K = horizon - 1
    for rnd in range(12):
        if rnd in (0, 5, 11):
            numer = model.numerators + model.gamma
            denom = model.denominators + 1
            numer[np.asarray(model.dataset.seen_indices, dtype=np.int64)] = -np.inf
            scores = numer / denom
            top = np.argsort(-scores, kind="stable")[:K + D].astype(np.int32)
            with np.errstate(invalid="ignore"):
                vals = LKNNModel._top_sum_host(numer, denom, scores, np.sort(nbr), K, top, block=700)
            ref = g[f"h{horizon}_values_r{rnd}"]
            assert np.array_equal(np.isnan(vals), np.isnan(ref))
            ok = ~np.isnan(ref)
            assert np.array_equal(vals[ok].view(np.uint64), ref[ok].view(np.uint64)), (horizon, rnd)
        model.condition_(int(picks[rnd]), int(truth[int(picks[rnd])]))


def test_incremental_order_equals_the_full_argsort():
    """VERDICT r3 "What's missing" #5: condition_() no longer re-sorts all N scores (the reference's np.argsort per
    answer, LKNN_model.py:185); the handful of changed nodes is merged back into the descending order.  Against the full
    argsort over many random updates, including moves to the very top and bottom and repeated nodes."""
    from seesaw_amd.loops.LKNN_model import _reinsert_sorted
    rng = np.random.default_rng(0)
    n = 5000
    score = rng.random(n)
    idx = np.argsort(-score)
    sc = score[idx]
    for step in range(300):
        m = int(rng.integers(1, 25))
        changed = rng.integers(0, n, size=m)                  # may repeat
        vals = rng.random(m) if step % 7 else rng.choice([-1.0, 2.0], size=m) + rng.random(m) * 1e-9
        score[changed] = vals                                  # (a repeated node ends with its last value, as numpy assigns)
        idx, sc = _reinsert_sorted(idx, sc, changed, score)
        want = np.argsort(-score)
        assert np.array_equal(idx, want), step
        assert np.array_equal(sc, score[want])
