"""GPU: BASELINE config C5 at its stated small size -- 1 109 images x 13 tiles = 14 417 vectors, 30 rounds, batch 1,
shortlist 50 -- whole sessions of plain / knn_prop2 / multi_reg / pseudo_lr against the REFERENCE's own Session +
benchmark_loop on the same synthetic dataset (tests/golden/c5_sequence.npz, oracle/gen_golden.py::gen_c5_sequence;
k-NN graph there: the reference's compute_exact_knn, here: ssw_knn_build).  Every round's image must equal the
reference's over the rounds on which the reference agrees with itself across its torch seeds."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
MATRIX = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
LP = dict(matrix_options=MATRIX, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4,
          prior_weight=1.0)
LOGREG = dict(class_weights=1.0, scale="centered", reg_lambda=1.0, max_iter=200.0, lr=1, fit_intercept=False)
OPTIONS = {
    "plain": None,
    "knn_prop2": LP,
    "multi_reg": dict(label_loss_type="ce_loss", rank_loss_margin=0.2, use_qvec_norm=None, reg_data_lambda=0.0,
                      reg_norm_lambda=100.0, reg_query_lambda=0.0, verbose=False, max_iter=200, pos_weight="balanced",
                      lr=1.0, matrix_options=MATRIX),
    "pseudo_lr": dict(switch_over=True, real_sample_weight=1.0, sample_size=10000, log_reg_params=LOGREG,
                      label_prop_params=LP),
}


@pytest.fixture(scope="module")
def c5():
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset
    g = np.load(os.path.join(GOLDEN, "c5_sequence.npz"))
    ds = make_dataset("lvis", knn_k=10, **json.loads(str(g["make"])))
    ds.embedding.noise = float(g["noise"])
    return g, GlobalDataManager().add(ds), ds


def _prefix(seqs):
    n = 0
    while n < min(len(x) for x in seqs) and all(x[n] == seqs[0][n] for x in seqs):
        n += 1
    return n


@pytest.mark.parametrize("name", ["plain", "knn_prop2", "multi_reg", "pseudo_lr"])
def test_c5_session_matches_reference(c5, name):
    import torch
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    g, gdm, ds = c5
    p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale", c_name=None), interactive=name,
                      interactive_options=OPTIONS[name], shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                      batch_size=1, start_policy="from_start" if name == "knn_prop2" else "after_first_batch",
                      index_options={"use_vec_index": False})
    b = BenchParams(name=name, ground_truth_category="c1", qstr="a c1", n_batches=30, max_results=10 ** 6)
    np.random.seed(0)
    torch.manual_seed(0)
    ret = make_session(gdm, p, b=b)
    boxes, _ = ds.load_ground_truth()
    out = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
    shown = np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1) for a in ret["session"].acc_indices])
    ref = g[f"{name}_shown"]
    seqs = [ref] + [g[f"{name}_shown_seed{s}"] for s in g["seeds"][1:] if f"{name}_shown_seed{s}" in g.files]
    stable = _prefix(seqs)
    same = _prefix([shown, ref])
    print(f"C5 {name}: reference reproduces itself over {stable} of {len(ref)} rounds; ours equals it over {same}")
    assert len(shown) == len(ref) == 30
    assert np.array_equal(shown[:stable], ref[:stable]), (shown.tolist(), ref.tolist())
    if stable == len(ref):
        assert out["nfound"] == int(g[f"{name}_nfound"])
    # the reference is seed-independent at this size for every loop but possibly the L-BFGS ones: hold a floor
    assert stable >= {"plain": 30, "knn_prop2": 30, "pseudo_lr": 30, "multi_reg": 19}[name], stable  # as recorded


def test_a_new_text_query_voids_the_shortlist_refine_selected(c5, monkeypatch):
    """Round 6: KnnProp2.refine propagates AND selects the next shortlist in one device call (ssw_labelprop_round); the
    next next_batch() uses it -- unless something it depends on changed in between.  A session that refines, then gets a
    NEW text query, then asks for the next batch must show what the three-call path (SSW_NO_FUSED_ROUND=1) shows, and so
    must a session that asks for two batches in a row."""
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.seesaw_bench import fill_imdata
    from seesaw_amd.seesaw_session import make_session
    g, gdm, ds = c5
    boxes, _ = ds.load_ground_truth()
    boxes = boxes.assign(description="a " + boxes.category.astype(str))
    box_c1 = boxes[boxes.category == "c1"]
    p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale", c_name=None), interactive="knn_prop2",
                      interactive_options=OPTIONS["knn_prop2"], shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                      batch_size=1, start_policy="after_first_batch", index_options={"use_vec_index": False})
    b = BenchParams(name="knn_prop2", ground_truth_category="c1", qstr="a c1", n_batches=30, max_results=10 ** 6)

    def run(fused):
        if fused:
            monkeypatch.delenv("SSW_NO_FUSED_ROUND", raising=False)
        else:
            monkeypatch.setenv("SSW_NO_FUSED_ROUND", "1")
        import contextlib
        import io
        shown = []
        with contextlib.redirect_stdout(io.StringIO()):
            s = make_session(gdm, p, b=b)["session"]
            s.set_text("a c1")

            def one_round(refine=True):
                shown.extend(int(v) for v in s.next())
                batch = [fill_imdata(im, box_c1, b) for im in s.last_batch()]
                s.update_last_batch(batch)
                if refine:
                    s.refine()
            for _ in range(5):
                one_round()
            s.set_text("a c0")               # a new query between refine() and next()
            for _ in range(3):
                one_round()
            one_round(refine=False)           # two batches in a row: the second finds no shortlist waiting
            one_round()
            one_round()
        return shown

    fused, plain = run(True), run(False)
    assert len(fused) == 11 and len(set(fused)) == 11
    assert fused == plain, (fused, plain)
