"""CPU: the `avg_score` aggregation (score_frame2 / box_join, multiscale_index.py:112-150) -- the numpy oracle and
the product's host fallback (`rescore_candidates`) against what the reference returned on the tile scores it formed
itself (tests/golden/multiscale_query.npz: avg_*_cand_rows / avg_*_cand_scores / avg_*_dbidxs / avg_*_activations)."""
import os

import numpy as np
import pandas as pd
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "multiscale_query.npz"))


@pytest.mark.parametrize("aug", ["all", "greater", "adjacent"])
def test_oracle_reproduces_reference_bit_for_bit(g, oracle, aug):
    m = g["pyr_meta"]
    rows, sc = g[f"avg_{aug}_cand_rows"], g[f"avg_{aug}_cand_scores"]
    ids, best, scores = oracle.rescore_avg_score(m[rows, 0].astype(np.int64), m[rows, 2:6].astype(np.float32),
                                                 m[rows, 1], sc, 10, aug)
    act = g[f"avg_{aug}_activations"]
    assert np.array_equal(ids, g[f"avg_{aug}_dbidxs"])
    assert np.array_equal(m[rows[best], 2:6], act[:, :4])                 # the same tile represents the image
    assert np.array_equal(scores.astype(np.float64), act[:, 5])           # f32 scores, bit for bit


@pytest.mark.parametrize("aug", ["all", "greater", "adjacent"])
def test_host_fallback_matches_reference(g, aug):
    from seesaw_amd.indices.multiscale.multiscale_index import rescore_candidates
    m = g["pyr_meta"]
    rows, sc = g[f"avg_{aug}_cand_rows"], g[f"avg_{aug}_cand_scores"]
    meta = pd.DataFrame({"dbidx": m[rows, 0].astype(np.int64), "zoom_level": m[rows, 1].astype(np.int16),
                         "x1": m[rows, 2].astype(np.float32), "y1": m[rows, 3].astype(np.float32),
                         "x2": m[rows, 4].astype(np.float32), "y2": m[rows, 5].astype(np.float32), "score": sc})
    res = rescore_candidates(meta, 10, agg_method="avg_score", aug_larger=aug)
    act = g[f"avg_{aug}_activations"]
    assert np.array_equal(res["dbidxs"], g[f"avg_{aug}_dbidxs"])
    got = np.stack([a[["x1", "y1", "x2", "y2", "dbidx", "score"]].values[0].astype(np.float64) for a in res["activations"]])
    assert np.array_equal(got[:, :5], act[:, :5])
    assert np.abs(got[:, 5] - act[:, 5]).max() <= 1e-6


def test_iou_dtype_follows_the_frames():
    """two float32 frames -> float32 IoU (what the reference's torch tensors give); a float64 side promotes"""
    from seesaw_amd.box_utils import box_iou
    a = pd.DataFrame({"x1": np.float32([0, 10]), "y1": np.float32([0, 10]), "x2": np.float32([224, 234]), "y2": np.float32([224, 100.3])})
    b = pd.DataFrame({"x1": [5.0], "y1": [5.0], "x2": [100.7], "y2": [50.1]})
    assert box_iou(a, a).dtype == np.float32 and box_iou(a, b).dtype == np.float64
    assert np.allclose(np.diag(box_iou(a, a)), 1.0)


@pytest.mark.parametrize("aug", ["all", "greater", "adjacent"])
def test_cont_weighted_host_form_matches_reference(aug):
    """aug_weight='cont_weighted' (softmax of containment over all joined partners, multiscale_index.py:133-145):
    the host form against the reference's per-image scores and returned images (tests/golden/contweighted.npz)"""
    from seesaw_amd.indices.multiscale.multiscale_index import rescore_candidates, score_frame2
    g = np.load(os.path.join(GOLDEN, "contweighted.npz"))
    m = g["pyr_meta"]
    rows, sc = g[f"cw_{aug}_cand_rows"], g[f"cw_{aug}_cand_scores"]
    meta = pd.DataFrame({"dbidx": m[rows, 0].astype(np.int64), "zoom_level": m[rows, 1].astype(np.int16),
                         "x1": m[rows, 2].astype(np.float32), "y1": m[rows, 3].astype(np.float32),
                         "x2": m[rows, 4].astype(np.float32), "y2": m[rows, 5].astype(np.float32), "score": sc})
    for dbidx, want in zip(g[f"cw_{aug}_frame_dbidx"], g[f"cw_{aug}_frame_score"]):
        tup = score_frame2(meta[meta.dbidx == dbidx], agg_method="avg_score", aug_larger=aug, aug_weight="cont_weighted")
        assert abs(float(tup.score.iloc[0]) - want) <= 1e-6, (dbidx, float(tup.score.iloc[0]), want)
    res = rescore_candidates(meta, 10, agg_method="avg_score", aug_larger=aug, aug_weight="cont_weighted")
    assert np.array_equal(res["dbidxs"], g[f"cw_{aug}_dbidxs"])
    got = np.stack([a[["x1", "y1", "x2", "y2", "dbidx", "score"]].values[0].astype(np.float64) for a in res["activations"]])
    act = g[f"cw_{aug}_activations"]
    assert np.array_equal(got[:, :5], act[:, :5]) and np.abs(got[:, 5] - act[:, 5]).max() <= 1e-6
