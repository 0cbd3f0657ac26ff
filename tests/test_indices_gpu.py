"""GPU parity of the reference-shaped indices (CoarseIndex / MultiscaleIndex / VectorIndex /
InteractiveQuery) against golden outputs captured from the reference itself
(tests/golden/scan_topk.npz, multiscale_query.npz; generator: oracle/gen_golden.py)."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _meta_from_tiles(tiles, dbidx_of_position, seed):
    # must mirror oracle/gen_golden.py::synth_vector_meta
    rng = np.random.default_rng(seed)
    return rng


def test_query_prelim_matches_reference(oracle):
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex
    g = np.load(os.path.join(GOLDEN, "scan_topk.npz"))
    for c in range(int(g["n_cases"])):
        seed, n_images, k = int(g[f"c{c}_seed"]), int(g[f"c{c}_n_images"]), int(g[f"c{c}_k"])
        tiles = g[f"c{c}_tiles"]
        dbidx_of_position = np.arange(n_images) * 3 + 5
        row_dbidx = np.repeat(dbidx_of_position, tiles)
        n = row_dbidx.shape[0]
        meta = pd.DataFrame({"dbidx": row_dbidx, "zoom_level": 0, "x1": 0.0, "y1": 0.0, "x2": 224.0, "y2": 224.0})
        X = oracle.synth_rows(seed, 0, n, 512)
        q = oracle.synth_query(seed)
        index = MultiscaleIndex(embedding=None, vectors=X, vector_meta=meta)
        exclude = BitMap(dbidx_of_position[g[f"c{c}_excl_pos"]])
        df = index._query_prelim(vector=q, topk_dbidx=k, exclude_dbidx=exclude, force_exact=True)
        ref_dbidx, ref_score = g[f"c{c}_dbidx"], g[f"c{c}_max_score"]
        assert df.shape[0] == ref_dbidx.shape[0]
        band = oracle.rounding_band(X, q)
        # identical image SET as the reference; order may differ only inside the rounding band
        assert set(df.dbidx.tolist()) == set(ref_dbidx.tolist())
        assert np.abs(df.max_score.values - ref_score).max() <= band or np.array_equal(df.dbidx.values, ref_dbidx)
        same_order = np.array_equal(df.dbidx.values, ref_dbidx)
        if not same_order:
            ref_of = dict(zip(ref_dbidx.tolist(), ref_score.tolist()))
            for a, b in zip(df.dbidx.values, ref_dbidx):
                assert abs(ref_of[int(a)] - ref_of[int(b)]) <= band
        # index.score() against the reference's head of `vectors @ q`
        assert np.abs(index.score(q)[:256] - g[f"c{c}_scores_head"]).max() <= band


def test_coarse_query_matches_reference(oracle):
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.coarse.coarse_index import CoarseIndex
    g = np.load(os.path.join(GOLDEN, "scan_topk.npz"))
    n, seed = int(g["coarse_n"]), int(g["coarse_seed"])
    X = oracle.synth_rows(seed, 0, n, 512)
    q = oracle.synth_query(seed)
    idx = CoarseIndex(embedding=None, vectors=X, vector_meta=pd.DataFrame({"dbidx": np.arange(n) * 2 + 1}))
    excl = BitMap(g["coarse_excl"])
    res = idx.query(topk=100, vector=q, exclude=excl)
    assert np.array_equal(res["dbidxs"], g["coarse_dbidxs"])
    assert res["nextstartk"] == int(g["coarse_nextstartk"])
    got = np.array([a.score.values[0] for a in res["activations"]], dtype=np.float32)
    assert np.abs(got - g["coarse_scores"]).max() <= oracle.rounding_band(X, q)
    assert list(res["activations"][0].columns) == ["x1", "y1", "x2", "y2", "dbidx", "score"]
    # exhausted index: the reference returns a pair of empty arrays (coarse_index.py:61-62)
    out = idx.query(topk=5, vector=q, exclude=BitMap(np.arange(n) * 2 + 1))
    assert isinstance(out, tuple) and out[0].shape == (0,)
    # getXy through the stateful query
    qq = idx.new_query()
    r = qq.query_stateful(vector=q, batch_size=10)
    assert len(qq.returned) == 10
    from seesaw_amd.basic_types import Box
    for i, d in enumerate(r["dbidxs"]):
        qq.label_db.put(int(d), [Box(x1=0, y1=0, x2=1, y2=1, marked_accepted=True)] if i % 2 == 0 else [])
    Xt, yt = qq.getXy()
    assert Xt.shape == (10, 512) and yt.sum() == 5
    pos, neg = qq.getXy(get_positions=True)
    assert sorted(np.concatenate([pos, neg]).tolist()) == sorted(((np.sort(r["dbidxs"]) - 1) // 2).tolist())


def test_multiscale_query_matches_reference(oracle):
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    m = g["meta"]
    meta = pd.DataFrame({"dbidx": m[:, 0].astype(np.int64), "zoom_level": m[:, 1].astype(np.int16),
                         "x1": m[:, 2].astype(np.float32), "y1": m[:, 3].astype(np.float32),
                         "x2": m[:, 4].astype(np.float32), "y2": m[:, 5].astype(np.float32)})
    seed = int(g["seed"])
    X = oracle.synth_rows(seed, 0, meta.shape[0], 512)
    q = oracle.synth_query(seed)
    band = oracle.rounding_band(X, q)
    index = MultiscaleIndex(embedding=None, vectors=X, vector_meta=meta)
    qq = index.new_query()
    for rnd in range(4):
        res = qq.query_stateful(vector=q, batch_size=5, shortlist_size=50, force_exact=True,
                                agg_method="plain_score", aug_larger="all",
                                rescore_method=lambda vecs: vecs @ q.reshape(-1, 1))
        assert np.array_equal(res["dbidxs"], g[f"r{rnd}_dbidxs"]), rnd
        acts = np.stack([a[["x1", "y1", "x2", "y2", "dbidx", "score"]].values[0].astype(np.float64)
                         for a in res["activations"]])
        ref = g[f"r{rnd}_activations"]
        assert np.array_equal(acts[:, :5], ref[:, :5])          # same best tile box per image
        assert np.abs(acts[:, 5] - ref[:, 5]).max() <= band
    assert len(qq.returned) == 20
    q2 = oracle.synth_query(seed + 1)
    res = index.query(vector=q, vector2=q2, topk=5, shortlist_size=50, exclude=BitMap(), force_exact=True,
                      agg_method="plain_score", aug_larger="all", rescore_method=None)
    assert np.array_equal(res["dbidxs"], g["v2_dbidxs"])
    got = np.array([a.score.values[0] for a in res["activations"]])
    assert np.abs(got - g["v2_scores"]).max() <= 2 * band


def _pyramid_index(g, oracle):
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex
    m = g["pyr_meta"]
    meta = pd.DataFrame({"dbidx": m[:, 0].astype(np.int64), "zoom_level": m[:, 1].astype(np.int16),
                         "x1": m[:, 2].astype(np.float32), "y1": m[:, 3].astype(np.float32),
                         "x2": m[:, 4].astype(np.float32), "y2": m[:, 5].astype(np.float32)})
    seed = int(g["pyr_seed"])
    X = oracle.synth_rows(seed, 0, meta.shape[0], 512)
    return MultiscaleIndex(embedding=None, vectors=X, vector_meta=meta), meta, X, oracle.synth_query(seed)


@pytest.mark.parametrize("aug", ["all", "greater", "adjacent"])
def test_avg_score_query_matches_reference(oracle, aug):
    """MultiscaleIndex.query(agg_method='avg_score') = the aggregation scripts/configs/std_bench.yaml uses, on a
    3-level tile pyramid: dbidxs and the activation boxes equal the reference's, scores to 1e-6 (the tile scores
    under the average come from the scan kernel instead of BLAS, hence not bit-identical)."""
    from seesaw_amd.bitmap import BitMap
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    index, meta, X, q = _pyramid_index(g, oracle)
    res = index.query(vector=q, topk=10, shortlist_size=50, exclude=BitMap(meta.dbidx.values[:40]), force_exact=True,
                      agg_method="avg_score", aug_larger=aug, rescore_method=None)
    ref = g[f"avg_{aug}_activations"]
    assert np.array_equal(res["dbidxs"], g[f"avg_{aug}_dbidxs"])
    acts = np.stack([a[["x1", "y1", "x2", "y2", "dbidx", "score"]].values[0].astype(np.float64) for a in res["activations"]])
    assert np.array_equal(acts[:, :5], ref[:, :5])
    assert np.abs(acts[:, 5] - ref[:, 5]).max() <= 1e-6
    assert res["activations"][0].score.dtype == np.float32


@pytest.mark.parametrize("aug", ["all", "greater", "adjacent"])
def test_cont_weighted_query_matches_reference(oracle, aug):
    """aug_weight='cont_weighted' on the device (k_avg_score's softmax-of-containment branch): the reference's images
    and activation boxes, scores to 2e-6 (f32 softmax + dot in partner order against scipy / BLAS); every candidate
    image's aggregated score against what the reference's score_frame2 returned for it"""
    from seesaw_amd.bitmap import BitMap
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    gc = np.load(os.path.join(GOLDEN, "contweighted.npz"))
    index, meta, X, q = _pyramid_index(g, oracle)
    res = index.query(vector=q, topk=10, shortlist_size=50, exclude=BitMap(meta.dbidx.values[:40]), force_exact=True,
                      agg_method="avg_score", aug_larger=aug, aug_weight="cont_weighted", rescore_method=None)
    ref = gc[f"cw_{aug}_activations"]
    assert np.array_equal(res["dbidxs"], gc[f"cw_{aug}_dbidxs"])
    acts = np.stack([a[["x1", "y1", "x2", "y2", "dbidx", "score"]].values[0].astype(np.float64) for a in res["activations"]])
    assert np.array_equal(acts[:, :5], ref[:, :5])
    assert np.abs(acts[:, 5] - ref[:, 5]).max() <= 2e-6
    pos = np.searchsorted(index._dbidx, gc[f"cw_{aug}_frame_dbidx"])
    index._dev.scores(q)
    scores, _ = index._dev.rescore_avg(pos, aug, aug_weight="cont_weighted")
    assert np.abs(scores.astype(np.float64) - gc[f"cw_{aug}_frame_score"]).max() <= 2e-6


def test_avg_score_vector2_matches_reference(oracle):
    from seesaw_amd.bitmap import BitMap
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    index, meta, X, q = _pyramid_index(g, oracle)
    q2 = oracle.synth_query(int(g["pyr_seed"]) + 1)
    res = index.query(vector=q, vector2=q2, topk=10, shortlist_size=50, exclude=BitMap(), force_exact=True,
                      agg_method="avg_score", aug_larger="greater", rescore_method=None)
    ref = g["avg_v2_activations"]
    assert np.array_equal(res["dbidxs"], g["avg_v2_dbidxs"])
    acts = np.stack([a[["x1", "y1", "x2", "y2", "dbidx", "score"]].values[0].astype(np.float64) for a in res["activations"]])
    assert np.array_equal(acts[:, :5], ref[:, :5])
    assert np.abs(acts[:, 5] - ref[:, 5]).max() <= 1e-6


@pytest.mark.parametrize("aug", ["all", "greater", "adjacent"])
def test_avg_score_kernel_bit_exact_vs_oracle_on_every_image(oracle, aug):
    """ssw_index_rescore_avg over ALL images of the index against the numpy oracle fed the device's own tile
    scores: same best tile, aggregated f32 score identical bit for bit (IoU and the Kahan mean are the
    reference's arithmetic, op for op)."""
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    index, meta, X, q = _pyramid_index(g, oracle)
    tile_scores = index._dev.scores(q)                      # leaves the scores resident
    n_images = index._dbidx.shape[0]
    scores, rows = index._dev.rescore_avg(np.arange(n_images), aug)
    boxes = meta[["x1", "y1", "x2", "y2"]].values.astype(np.float32)
    zoom = meta.zoom_level.values
    for p in range(n_images):
        a, b = index._row_start[p], index._row_start[p + 1]
        j, sc, _ = oracle.avg_score_image(boxes[a:b], zoom[a:b], tile_scores[a:b], aug)
        assert rows[p] == a + j, (p, rows[p], a + j)
        assert np.float32(sc).view(np.uint32) == scores[p:p + 1].view(np.uint32)[0], (p, sc, scores[p])
    # the vector2 form subtracts a second score per tile before aggregating
    minus = np.linspace(-0.01, 0.01, X.shape[0]).astype(np.float32)
    scores2, rows2 = index._dev.rescore_avg(np.arange(n_images), aug, minus)
    for p in (0, 17, n_images - 1):
        a, b = index._row_start[p], index._row_start[p + 1]
        j, sc, _ = oracle.avg_score_image(boxes[a:b], zoom[a:b], tile_scores[a:b] - minus[a:b], aug)
        assert rows2[p] == a + j and np.float32(sc).view(np.uint32) == scores2[p:p + 1].view(np.uint32)[0]


def test_avg_score_host_and_device_paths_agree(oracle):
    """float64 tile boxes keep the reference's float64 IoU on the host; float32 boxes run on the device:
    same images either way on this data"""
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    index, meta, X, q = _pyramid_index(g, oracle)
    meta64 = meta.assign(**{c: meta[c].astype(np.float64) for c in ("x1", "y1", "x2", "y2")})
    host = MultiscaleIndex(embedding=None, vectors=X, vector_meta=meta64)
    assert index._has_tile_meta and not host._has_tile_meta
    kw = dict(vector=q, topk=10, shortlist_size=50, exclude=BitMap(), force_exact=True, agg_method="avg_score",
              aug_larger="all", rescore_method=None)
    a, b = index.query(**kw), host.query(**kw)
    assert np.array_equal(a["dbidxs"], b["dbidxs"])


def test_vector_index(oracle, tmp_path):
    from seesaw_amd.vector_index import VectorIndex, build_annoy_idx
    X = oracle.synth_rows(1, 0, 5000, 512)
    q = oracle.synth_query(5)
    path = str(tmp_path / "vectors.annoy")
    assert build_annoy_idx(vecs=X, output_path=path, n_trees=10) >= 0
    vi = VectorIndex(load_path=path)
    assert vi.ready()
    ids, scores = vi.query(q.reshape(1, 512), top_k=200)
    ref = oracle.topk_images_tiebreak(oracle.scores_kernel_order(X, q), None, 5000, [], 200)
    assert np.array_equal(ids, ref[0]) and np.array_equal(scores, ref[1])
    with pytest.raises(AssertionError):
        vi.query(q[:100], top_k=5)


def test_multiscale_getxy_and_subset(oracle):
    from seesaw_amd.basic_types import Box
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex
    n_images = 50
    dbidx = np.repeat(np.arange(n_images), 4)
    boxes = np.tile(np.array([[0, 0, 224, 224], [224, 0, 448, 224], [0, 224, 224, 448], [0, 0, 448, 448]], np.float32), (n_images, 1))
    meta = pd.DataFrame({"dbidx": dbidx, "zoom_level": np.tile([0, 0, 0, 1], n_images).astype(np.int16),
                         "x1": boxes[:, 0], "y1": boxes[:, 1], "x2": boxes[:, 2], "y2": boxes[:, 3]})
    X = oracle.synth_rows(3, 0, meta.shape[0], 512)
    index = MultiscaleIndex(embedding=None, vectors=X, vector_meta=meta)
    qq = index.new_query()
    qq.label_db.put(7, [Box(x1=300, y1=10, x2=400, y2=100, marked_accepted=True)])  # overlaps tiles 1 and 3
    qq.label_db.put(9, [])
    df = qq.getXy()
    assert list(df.columns) == ["dbidx", "ys", "max_iou"]
    assert df.index.tolist() == [28, 29, 30, 31, 36, 37, 38, 39]
    assert df.ys.tolist() == [0, 1, 0, 1, 0, 0, 0, 0]
    pos, neg = qq.getXy(get_positions=True)
    assert pos.tolist() == [29, 31] and len(neg) == 6
    sub = index.subset(BitMap([3, 4, 5]))
    assert len(sub) == 3 and sub.vectors.shape[0] == 12
    assert index.subset(BitMap(range(n_images))) is index


def test_avg_score_over_device_f64_scores_equals_the_host_loop():
    """ssw_index_rescore_avg_f64 (float64 scores on the device: what the graph loops re-score with) against the host
    restatement of the reference's rescore_candidates over the same float64 column: same images, same activation
    boxes, bit-identical scores -- for the three aug_larger modes"""
    import ctypes
    from seesaw_amd import _lib
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex, rescore_candidates
    from seesaw_amd.synthetic import make_dataset
    ds = make_dataset("lvis", n_images=120, tiles_per_image=13, n_categories=2, positive_frac=0.1, seed=3, knn_k=0)
    idx = MultiscaleIndex(embedding=ds.embedding, vectors=ds.vectors, vector_meta=ds.vector_meta, vec_index=None)
    assert idx._has_tile_meta
    rng = np.random.default_rng(9)
    scores64 = rng.uniform(0, 1, ds.vectors.shape[0])  # float64, like a label-propagation result
    scores64[rng.integers(0, scores64.shape[0], 200)] = 0.5  # exact ties
    lib = _lib.load()
    torch = pytest.importorskip("torch")
    dev_scores = torch.from_numpy(scores64).cuda()
    positions = np.sort(rng.choice(120, size=50, replace=False)).astype(np.int64)
    cand = pd.DataFrame({"dbidx": idx._dbidx[positions], "max_score": np.zeros(50)})
    cand.attrs["positions"] = positions
    for aug in ("all", "greater", "adjacent"):
        got = idx.rescore_avg_from_device_scores(cand, 10, aug, dev_scores.data_ptr())
        rows = idx._candidate_rows(positions)
        want = rescore_candidates(idx.vector_meta.iloc[rows].assign(score=scores64[rows]), 10, agg_method="avg_score",
                                  aug_larger=aug)
        assert np.array_equal(got["dbidxs"], want["dbidxs"]), aug
        recs = got["activations"].records()
        for i, frame in enumerate(want["activations"]):
            ref = frame[["x1", "y1", "x2", "y2", "score"]].to_numpy(dtype=np.float64)[0]
            assert np.array_equal(np.asarray(recs[i], dtype=np.float64)[:4], ref[:4]), (aug, i)
            assert recs[i][4] == ref[4], (aug, i, recs[i][4], ref[4])
    idx._dev.close()
