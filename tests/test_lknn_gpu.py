"""GPU: the vectorised two-step look-ahead of L-KNN active search (ssw_lknn_top_sum) inside
efficient_nonmyopic_search(implementation='vectorized'): over a 12-round planning session on a random 10-regular
graph the node picked and its value equal the reference's every round, and the full value vectors are bit-identical
(tests/golden/lknn.npz; horizons 2, 9, 20 and 101 = K of 1, 8, 19 and 100 future picks).  Plus the recursive
definition ('loop') against the vectorised form, and the two loops behind the session API."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from test_lknn_cpu import session_graph

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("horizon", [2, 9, 20, 101])
def test_planning_session_matches_reference(horizon):
    from seesaw_amd.loops.LKNN_model import LKNNModel, initial_gamma_array
    from seesaw_amd.research.active_search.common import Dataset
    from seesaw_amd.research.active_search.efficient_nonmyopic_search import efficient_nonmyopic_search
    g = np.load(os.path.join(GOLDEN, "lknn.npz"))
    N, D, nbr, W, truth = session_graph(g)
    model = LKNNModel.from_dataset(Dataset.from_vectors(np.zeros((N, 1))), weight_matrix=W, gamma=initial_gamma_array(0.1, N))
    for rnd in range(12):
        res = efficient_nonmyopic_search(model, reward_horizon=horizon, lookahead_limit=2, pruning_on=False,
                                         implementation="vectorized")
        assert int(res.index) == int(g[f"h{horizon}_picks"][rnd]), (horizon, rnd)
        assert res.value == g[f"h{horizon}_values"][rnd], (horizon, rnd, res.value, g[f"h{horizon}_values"][rnd])
        if rnd in (0, 5, 11):
            _, _, vals = model.top_sum(K=horizon - 1, return_values=True)
            ref = g[f"h{horizon}_values_r{rnd}"]
            assert np.array_equal(np.isnan(vals), np.isnan(ref))
            ok = ~np.isnan(ref)
            assert np.array_equal(vals[ok].view(np.uint64), ref[ok].view(np.uint64)), (horizon, rnd)
        model.condition_(int(res.index), int(truth[int(res.index)]))


def test_vectorised_equals_recursive_definition_on_a_small_graph():
    """'loop' = the recursive two-step look-ahead over every remaining node; 'vectorized' = the shared-list form on
    the GPU: same pick, values equal to rounding, over several rounds"""
    from seesaw_amd.loops.LKNN_model import LKNNModel, initial_gamma_array
    from seesaw_amd.research.active_search.common import Dataset
    from seesaw_amd.research.active_search.efficient_nonmyopic_search import efficient_nonmyopic_search
    N, D = 60, 4
    rng = np.random.default_rng(5)
    nbr = np.stack([rng.choice(N, D, replace=False) for _ in range(N)]).astype(np.int32)
    W = sp.csr_array((np.ones(N * D), nbr.reshape(-1), np.arange(0, N * D + 1, D)), shape=(N, N))
    gamma = np.clip(initial_gamma_array(0.2, N) + rng.uniform(-0.1, 0.1, N), 0.01, 0.9)
    model = LKNNModel.from_dataset(Dataset.from_vectors(np.zeros((N, 1))), weight_matrix=W, gamma=gamma)
    for rnd in range(5):
        a = efficient_nonmyopic_search(model, reward_horizon=6, lookahead_limit=2, pruning_on=False, implementation="vectorized")
        b = efficient_nonmyopic_search(model, reward_horizon=6, lookahead_limit=2, pruning_on=False, implementation="loop")
        assert int(a.index) == int(b.index) and abs(a.value - b.value) < 1e-12, (rnd, a.index, b.index, a.value, b.value)
        model.condition_(int(a.index), int(rng.random() < 0.3))


@pytest.mark.parametrize("loop", ["lknn", "active_search"])
def test_active_search_loops_through_the_session(loop):
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset
    ds = make_dataset("lvis", n_images=1500, tiles_per_image=1, n_categories=2, positive_frac=0.03, seed=4, knn_k=10, signal=0.5)
    gdm = GlobalDataManager().add(ds)
    matrix = dict(knn_path="nndescent60", symmetric=False, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
    lp = dict(matrix_options=matrix, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4, prior_weight=1.0)
    if loop == "lknn":
        opts = dict(gamma=0.1, use_clip_as_gamma=False, **lp)
    else:
        opts = dict(gamma=dict(mode="clip", calibration="sigmoid", a=10.0, b=-0.2), reward_horizon=10, adjust_horizon=False,
                    max_steps=100, pruning_on=False, implementation="vectorized", **lp)
    p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="coarse"), interactive=loop, interactive_options=opts,
                      batch_size=1, shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                      start_policy="from_start", index_options={"use_vec_index": False})
    b = BenchParams(name=loop, ground_truth_category="c1", qstr="a c1", n_batches=12, max_results=10 ** 6)
    ret = make_session(gdm, p, b=b)
    boxes, _ = ds.load_ground_truth()
    out = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
    assert out["nseen"] == 12 and len(set(int(a[0]) for a in ret["session"].acc_indices)) == 12
