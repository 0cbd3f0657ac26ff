"""GPU end-to-end: Session + benchmark_loop over the synthetic datasets for every registered
feedback loop (BASELINE configs C1 and C5-shape).  Checks the invariants the reference's
benchmark_loop asserts (every id inside the subset, never repeated: seesaw_bench.py:316-320),
that feedback helps, and the summary round trip of BenchRunner.run_loop."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(interactive, options=None, batch_size=1, start_policy="after_first_batch"):
    from seesaw_amd.basic_types import IndexSpec, SessionParams
    return SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale", c_name=None), interactive=interactive,
                         interactive_options=options, batch_size=batch_size, shortlist_size=50,
                         agg_method="plain_score", aug_larger="greater", start_policy=start_policy,
                         index_options={"use_vec_index": False})


MATRIX = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
LOOPS = {
    "plain": None,
    "rocchio_update": dict(rocchio_alpha=1.0, rocchio_beta=0.5, rocchio_gamma=0.25, verbose=False),
    "multi_reg": dict(label_loss_type="ce_loss", rank_loss_margin=0.2, use_qvec_norm=None, reg_data_lambda=0.0,
                      reg_norm_lambda=100.0, reg_query_lambda=0.0, verbose=False, max_iter=200, pos_weight="balanced",
                      lr=1.0, matrix_options=MATRIX),
    "multi_reg_data": dict(label_loss_type="pairwise_rank_loss", rank_loss_margin=0.2, use_qvec_norm=None,
                           reg_data_lambda=1000.0, reg_norm_lambda=100.0, reg_query_lambda=10.0, verbose=False,
                           max_iter=100, pos_weight="balanced", lr=1.0, matrix_options=MATRIX),
    "knn_prop2": dict(matrix_options=MATRIX, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0,
                      calib_b=-0.4, prior_weight=1.0),
    "pseudo_lr": dict(switch_over=True, real_sample_weight=1.0, sample_size=2000,
                      log_reg_params=dict(class_weights=1.0, scale="centered", reg_lambda=1.0, max_iter=200.0, lr=1,
                                          fit_intercept=False),
                      label_prop_params=dict(matrix_options=MATRIX, normalize_scores=False,
                                             sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4, prior_weight=1.0)),
    "log_reg2": dict(class_weights=1.0, scale="centered", reg_lambda=1.0, max_iter=200.0, lr=1, fit_intercept=False),
}


@pytest.fixture(scope="module")
def lvis():
    from seesaw_amd.synthetic import GlobalDataManager, make_lvis_shape
    ds = make_lvis_shape(n_images=600, seed=3, knn_k=10)
    return GlobalDataManager().add(ds), ds


@pytest.mark.parametrize("name", list(LOOPS))
def test_benchmark_loop_all_loops(lvis, name):
    from seesaw_amd.basic_types import BenchParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    gdm, ds = lvis
    interactive = "multi_reg" if name.startswith("multi_reg") else name
    p = _params(interactive, LOOPS[name])
    b = BenchParams(name=name, ground_truth_category="c1", qstr="a c1", n_batches=25, max_results=5)
    np.random.seed(0)
    ret = make_session(gdm, p, b=b)
    boxes, qgt = ds.load_ground_truth()
    out = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
    assert 0 < out["nseen"] <= 25 and out["nfound"] >= 1
    assert len(out["latencies"]) in (out["nseen"], out["nseen"] - 1)
    state = ret["session"].get_state()
    shown = [im.dbidx for batch in state.gdata for im in batch]
    assert len(shown) == len(set(shown)) == out["nseen"]
    # the simulated user's verdicts agree with the ground truth
    positives = set(boxes[boxes.category == "c1"].dbidx.tolist())
    from seesaw_amd.basic_types import is_image_accepted
    for batch in state.gdata:
        for im in batch:
            assert is_image_accepted(im) == (im.dbidx in positives)


def test_c1_plain_top100(tmp_path):
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.seesaw_bench import BenchRunner, add_stats, get_all_session_summaries
    from seesaw_amd.synthetic import GlobalDataManager, make_c1
    gdm = GlobalDataManager().add(make_c1())
    p = SessionParams(index_spec=IndexSpec(d_name="c1", i_name="coarse"), interactive="plain", batch_size=100,
                      shortlist_size=100, agg_method="plain_score", start_policy="from_start")
    b = BenchParams(name="baseline", ground_truth_category="c0", qstr="a c0", n_batches=3, max_results=100)
    runner = BenchRunner(None, str(tmp_path), redirect_output=True, gdm=gdm)
    out_dir = runner.run_loop(b, p)
    summary = json.load(open(os.path.join(out_dir, "summary.json")))
    res = summary["result"]
    assert res["nimages"] == 10000 and res["ntotal"] == 100
    assert res["run_info"]["nseen"] == 300 and res["run_info"]["nfound"] > 30
    assert len(res["latencies"]) == 2
    df = add_stats(get_all_session_summaries(str(tmp_path), force_recompute=True))
    assert df.shape[0] == 1 and df.nseen.iloc[0] == 300 and 0 < df.average_precision.iloc[0] <= 1


def test_feedback_beats_plain(lvis):
    """on a hard query (text vector only weakly aligned) the multi_reg loop finds the
    positives in about as few batches as no feedback, or fewer."""
    from seesaw_amd.basic_types import BenchParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    gdm, ds = lvis
    ds.embedding.noise = 1.6
    ds.embedding.string_cache.clear()
    seen = {}
    try:
        for name in ("plain", "multi_reg"):
            p = _params(name, LOOPS[name])
            b = BenchParams(name=name, ground_truth_category="c2", qstr="a c2", n_batches=60, max_results=6)
            ret = make_session(gdm, p, b=b)
            boxes, _ = ds.load_ground_truth()
            out = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
            seen[name] = (out["nseen"], out["nfound"])
    finally:
        ds.embedding.noise = 0.35
        ds.embedding.string_cache.clear()
    assert seen["multi_reg"][1] >= seen["plain"][1]
    assert seen["multi_reg"][0] <= seen["plain"][0] + 3


GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


SEQUENCE_VARIANTS = ["plain", "rocchio_update", "multi_reg", "multi_reg_data", "knn_prop2", "plain_c", "log_reg2_c",
                     "plain_b", "knn_prop2_b", "pseudo_lr_b", "plain_avg", "knn_prop2_b_avg", "lknn_d", "active_search_d"]

# the active-search loops of the sequence fixture (row f-4 end to end; options as oracle/gen_golden.py wrote them)
_LP = dict(matrix_options=MATRIX, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4,
           prior_weight=1.0)
SEQUENCE_LOOPS = {
    "lknn": dict(gamma=0.1, use_clip_as_gamma=False, **_LP),
    "active_search": dict(gamma=dict(mode="clip", calibration="sigmoid", a=10.0, b=-0.2), reward_horizon=5,
                          adjust_horizon=True, max_steps=25, pruning_on=False, implementation="vectorized",
                          **{**_LP, "matrix_options": {**MATRIX, "symmetric": False}}),
}


@pytest.fixture(scope="module")
def sequence_datasets():
    """the synthetic datasets of tests/golden/bench_loop.npz, rebuilt from the parameters stored in it"""
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    specs = json.loads(str(g["datasets"]))
    assert int(g["knn_pool"]) == 11  # synthetic.knn_graph stores knn_k + 1 neighbours, as the generator did
    out = {}
    for key, spec in specs.items():
        coarse = spec["make"]["tiles_per_image"] == 1
        ds = make_dataset("lvis", knn_k=10 if (not coarse or spec.get("graph")) else 0, **spec["make"])
        ds.embedding.noise = spec["noise"]
        out[key] = (GlobalDataManager().add(ds), ds, coarse)
    return g, out


@pytest.mark.parametrize("name", SEQUENCE_VARIANTS)
def test_benchmark_loop_sequence_matches_reference(sequence_datasets, name):
    """The dbidx returned in every round, nfound and nseen equal what the REFERENCE's own
    Session + benchmark_loop + MultiscaleIndex / CoarseIndex + loops produced on the same synthetic
    datasets (tests/golden/bench_loop.npz, captured by oracle/gen_golden.py::gen_bench_loop with the
    reference's own compute_exact_knn graph).  Dataset B is the one where label propagation matters:
    knn_prop2_b and pseudo_lr_b differ from plain_b."""
    import torch
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    g, datasets = sequence_datasets
    names = [str(x) for x in g["names"]]
    assert sorted(names) == sorted(SEQUENCE_VARIANTS)
    i = names.index(name)
    key, interactive = str(g["variant_dataset"][i]), str(g["variant_interactive"][i])
    gdm, ds, coarse = datasets[key]
    opts_name = {"multi_reg_data": "multi_reg_data"}.get(name, interactive)
    over = dict(agg_method="plain_score", aug_larger="greater", batch_size=1)
    over.update(json.loads(str(g["session_overrides"])).get(name, {}))  # plain_avg / knn_prop2_b_avg: avg_score sessions
    p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="coarse" if coarse else "multiscale", c_name=None),
                      interactive=interactive, interactive_options={**LOOPS, **SEQUENCE_LOOPS}[opts_name], shortlist_size=50,
                      **over,
                      start_policy="from_start" if (interactive == "knn_prop2" or coarse) else "after_first_batch",
                      index_options={"use_vec_index": False})
    b = BenchParams(name=name, ground_truth_category="c1", qstr="a c1", n_batches=25, max_results=10)
    np.random.seed(0)
    torch.manual_seed(0)
    ret = make_session(gdm, p, b=b)
    boxes, _ = ds.load_ground_truth()
    out = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
    shown = np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1) for a in ret["session"].acc_indices])
    ref = g[f"{name}_shown"]
    stable = _stable_rounds(g, name)
    print(f"{name}: reference reproduces itself over {stable} of {len(ref)} rounds; ours equals it over "
          f"{_common_prefix([shown, ref])}")
    assert np.array_equal(shown[:stable], ref[:stable]), (shown.tolist(), ref.tolist())
    if stable == len(ref):
        assert np.array_equal(shown, ref)
        assert out["nfound"] == int(g[f"{name}_nfound"]) and out["nseen"] == int(g[f"{name}_nseen"])


def test_pseudo_lr_with_the_propagation_beside_the_draw(sequence_datasets, monkeypatch):
    """PseudoLR.refine on graphs of 2^18+ vectors records the labels, then runs the propagation on a helper thread
    while the main thread makes the pseudo-label draw (loops/pseudo_lr.py).  Forced here on the fixture's dataset B:
    the session must return the reference's images exactly as the serial order of calls does."""
    monkeypatch.setenv("SSW_PSEUDOLR_OVERLAP_FROM", "0")
    test_benchmark_loop_sequence_matches_reference(sequence_datasets, "pseudo_lr_b")
    from seesaw_amd.loops import pseudo_lr
    assert pseudo_lr._SIDE is not None, "the helper thread never ran"


def _common_prefix(seqs):
    n = 0
    while n < min(len(x) for x in seqs) and all(x[n] == seqs[0][n] for x in seqs):
        n += 1
    return n


def _stable_rounds(g, name):
    """rounds over which the REFERENCE returns the same images under all of the torch seeds the fixture was
    recorded with (its L-BFGS fits depend on the DataLoader shuffle; for `multi_reg` with ce_loss and no anchoring
    regulariser its own fits differ by 1e-3 .. 3e-2 in rank scores between seeds and the sequences part after 2
    rounds -- no implementation can be held to a sequence the reference does not reproduce itself)."""
    seqs = [g[f"{name}_shown"]] + [g[f"{name}_shown_seed{s}"] for s in g["seeds"][1:] if f"{name}_shown_seed{s}" in g.files]
    return _common_prefix(seqs)


def test_reference_sequence_stability_recorded():
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    assert _stable_rounds(g, "multi_reg_data") == len(g["multi_reg_data_shown"]) == 24
    assert _stable_rounds(g, "pseudo_lr_b") == len(g["pseudo_lr_b_shown"]) == 10
    assert _stable_rounds(g, "log_reg2_c") >= 10
    assert _stable_rounds(g, "plain") == 25


def test_multireg_session_vs_shuffle_free_reference(sequence_datasets):
    """`multi_reg` (ce_loss, no anchoring regulariser) is the variant whose reference sequences part after 2 rounds
    between torch seeds.  tests/golden/multireg_det.npz holds the reference's session with its DataLoader shuffle switched
    off -- ONE deterministic run, rows in storage order as here.  The objective is under-determined (tens of rows, 512
    dimensions), so f32 summation order still decides late rounds; the first rounds are held with a numeric floor."""
    import torch
    from seesaw_amd.basic_types import BenchParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    _, datasets = sequence_datasets
    gd = np.load(os.path.join(GOLDEN, "multireg_det.npz"))
    gdm, ds, _ = datasets["A"]
    p = _params("multi_reg", LOOPS["multi_reg"])
    b = BenchParams(name="multi_reg", ground_truth_category="c1", qstr="a c1", n_batches=25, max_results=10)
    np.random.seed(0)
    torch.manual_seed(0)
    ret = make_session(gdm, p, b=b)
    boxes, _ = ds.load_ground_truth()
    benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
    shown = np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1) for a in ret["session"].acc_indices])
    ref = gd["multi_reg_shown"]
    same = _common_prefix([shown, ref])
    print(f"multi_reg vs the shuffle-free reference session: identical over {same} of {len(ref)} rounds; "
          f"ours {shown.tolist()} reference {ref.tolist()}")
    assert same >= MULTIREG_DET_FLOOR, (same, shown.tolist(), ref.tolist())


MULTIREG_DET_FLOOR = 25  # measured (round 3): all 25 rounds equal the reference's shuffle-free session


def test_multireg_session_fits_against_reference():
    """the fits the reference's own session performed round by round (inputs captured in bench_loop.npz):
    loss and gradient agree at 1e-4 along its closure trajectory; the fitted direction is compared in rank
    scores of the labelled rows (printed; the reference's seeds themselves sit 1e-3 .. 3e-2 apart here)"""
    from seesaw_amd import _lib
    from seesaw_amd.feedback import FeedbackEngine
    from seesaw_amd.synthetic import make_dataset
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    ds = make_dataset("lvis", knn_k=0, **json.loads(str(g["datasets"]))["A"]["make"])
    eng = FeedbackEngine(512)
    obj = _lib.FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=0, fit_intercept=0, reg_kind=0, pos_weight=-1.0,
                           reg_weight=0.0, margin=0.2, reg_norm_lambda=100.0, reg_data_lambda=0.0, reg_query_lambda=0.0)
    checked = 0
    for r in range(int(g["multi_reg_n_fits"])):
        rows, y, img, q = (g[f"multi_reg_fit{r}_{k}"] for k in ("rows", "y", "img", "q"))
        _, inv, counts = np.unique(img, return_inverse=True, return_counts=True)
        eng.set_data(ds.vectors[rows], center=True)
        eng.set_targets(y, 1.0 / counts[inv])
        eng.set_query(q)
        if f"multi_reg_fit{r}_traj_w" in g.files:
            W, L, G = (g[f"multi_reg_fit{r}_traj_{k}"] for k in ("w", "loss", "grad"))
            for t in range(W.shape[0]):
                loss, grad, _ = eng.lossgrad(obj, W[t])
                assert abs(loss - L[t]) <= 1e-4 * max(1.0, abs(L[t])), (r, t, loss, L[t])
                assert np.abs(grad - G[t]).max() <= 1e-4 * max(1.0, np.abs(G[t]).max()), (r, t)
                checked += 1
    assert checked > 150


def test_sequence_fixture_discriminates_label_propagation():
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    assert not np.array_equal(g["knn_prop2_b_shown"], g["plain_b_shown"])
    assert not np.array_equal(g["pseudo_lr_b_shown"], g["plain_b_shown"])
    assert not np.array_equal(g["log_reg2_c_shown"], g["plain_c_shown"])


def test_readme_session_snippet():
    """the end-to-end snippet of README.md, as written there"""
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset
    from seesaw_amd.basic_types import IndexSpec, SessionParams
    from seesaw_amd.seesaw_session import make_session
    ds = make_dataset("demo", n_images=300, tiles_per_image=13, n_categories=2, positive_frac=0.05, seed=11, knn_k=10)
    p = SessionParams(index_spec=IndexSpec(d_name="demo", i_name="multiscale"), interactive="multi_reg",
                      interactive_options=dict(label_loss_type="ce_loss", rank_loss_margin=0.2, use_qvec_norm=None,
                                               reg_data_lambda=0.0, reg_norm_lambda=100.0, reg_query_lambda=0.0,
                                               verbose=False, max_iter=200, pos_weight="balanced", lr=1.0,
                                               matrix_options=None),
                      batch_size=1, shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                      start_policy="after_first_batch", index_options={"use_vec_index": False})
    session = make_session(GlobalDataManager().add(ds), p)["session"]
    session.set_text("a c1")
    shown = session.next()
    assert len(shown) == 1
    state = session.get_state()
    session.update_state(state)
    session.refine()
    again = session.next()
    assert len(again) == 1 and int(again[0]) != int(shown[0])
