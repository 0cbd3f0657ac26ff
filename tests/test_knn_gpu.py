"""GPU parity of the exact k-NN graph build (ssw_knn_build through DeviceIndex.knn /
seesaw_amd.knn_graph.compute_exact_knn) against the CPU oracle (bit-exact: neighbour ids, their order
and the f32 score bits) and against the reference's compute_exact_knn output (tests/golden/labelprop.npz,
BLAS summation there, so near-ties may swap within the rounding band)."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _check_vs_oracle(oracle, X, k, rows=None):
    from seesaw_amd.device_index import DeviceIndex
    dev = DeviceIndex.from_numpy(X)
    dst, score, redone = dev.knn(k)
    dev.close()
    if rows is None:
        o_dst, o_score = oracle.exact_knn(X, k)
        assert np.array_equal(dst, o_dst), np.nonzero((dst != o_dst).any(axis=1))[0][:10]
        assert np.array_equal(bits(score), bits(o_score))
    else:
        ids = np.arange(X.shape[0])
        for r in rows:
            s = oracle.scores_kernel_order(X, X[r])
            order = np.lexsort((ids, -s.astype(np.float64)))[:k + 1]
            assert np.array_equal(dst[r], order), r
            assert np.array_equal(bits(score[r]), bits(s[order])), r
    return redone


@pytest.mark.parametrize("n,k,dim", [(1, 1, 512), (5, 4, 512), (40, 10, 512), (130, 15, 256), (1500, 10, 512),
                                     (3000, 10, 512), (2500, 3, 768), (2000, 31, 512), (1800, 16, 512)])
def test_knn_bit_exact_vs_oracle(oracle, n, k, dim):
    k = min(k, n - 1) if n > 1 else 0
    X = oracle.synth_rows(100 + n, 0, n, dim)
    if k == 0:
        pytest.skip("a single vertex has no neighbours")
    redone = _check_vs_oracle(oracle, X, k)
    assert redone <= max(2, n // 50)


def test_knn_clustered_and_duplicate_rows(oracle):
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    X = g["X"].copy()  # clustered unit rows
    X[100:150] = X[7]  # 51 identical rows: exact score ties, ordered by row id
    redone = _check_vs_oracle(oracle, X, 10)
    assert redone <= X.shape[0] // 10


def test_compute_exact_knn_vs_reference_golden(oracle):
    from seesaw_amd.knn_graph import compute_exact_knn
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    df = compute_exact_knn(g["X"], int(g["k"]))
    assert df.shape[0] == g["src"].shape[0]
    assert np.array_equal(df.src_vertex.values, g["src"]) and np.array_equal(df.dst_rank.values, g["rank"])
    assert df.dst_vertex.dtype == np.int32 and df.distance.dtype == np.float32
    assert np.allclose(df.distance.values, g["dist"], rtol=0, atol=1e-6)  # f32 sums in a different order
    differs = df.dst_vertex.values != g["dst"]
    assert np.all(np.abs(df.distance.values[differs] - g["dist"][differs]) <= 1e-6) and differs.mean() < 0.01


@pytest.mark.parametrize("batched", [False, True])
def test_knn_two_hundred_thousand_rows_spot_checked(oracle, monkeypatch, batched):
    """three column levels; the symmetric all-rows-at-once path and the batched rectangular path (two row
    batches, what a graph too large for HBM-resident buffers takes); 68 rows checked bit for bit against
    full CPU scans"""
    if batched:
        monkeypatch.setenv("SSW_KNN_FORCE_BATCHED", "1")
    n, k = 200_000, 10
    X = oracle.synth_rows(77, 0, n, 512)
    rows = np.random.default_rng(3).integers(0, n, size=64).tolist() + [0, n - 1, 131071, 131072]
    redone = _check_vs_oracle(oracle, X, k, rows=rows)
    assert redone <= n // 100, redone


def test_build_knn_graph_saved_where_the_loops_look(oracle, tmp_path):
    """build_knn_graph writes <index>/knn_graph/<name>/forward.parquet; KNNGraph.from_file reads it back and
    the weight-matrix lookup of the loops (graph_based.py:28-66) finds it by name"""
    from seesaw_amd.indices.coarse.coarse_index import CoarseIndex
    from seesaw_amd.knn_graph import KNNGraph, build_knn_graph, compute_knn_from_nndescent
    from seesaw_amd.loops.graph_based import get_weight_matrix_from_index
    n = 700
    X = oracle.synth_rows(5, 0, n, 512)
    meta = pd.DataFrame({"dbidx": np.arange(n)})
    idx = CoarseIndex(embedding=None, vectors=X, vector_meta=meta, path=str(tmp_path))
    g = build_knn_graph(idx, name="exact10", n_neighbors=10)
    back = KNNGraph.from_file(idx.get_knng_path(name="exact10"))
    assert back.knn_df.equals(g.knn_df) and back.k == 10 and back.nvecs == n
    dst, score = oracle.exact_knn(X, 10)
    got = back.knn_df[back.knn_df.dst_rank > 0].sort_values(["src_vertex", "dst_rank"]).dst_vertex.values.reshape(n, 10)
    assert np.array_equal(got, dst[:, 1:])  # self edge is rank 0, the oracle lists the vertex itself first
    W = get_weight_matrix_from_index(idx, dict(knn_path="exact10", knn_k=10, edist=0.1, self_edges=False,
                                               normalized_weights=False, symmetric=True))
    assert W.shape == (n, n) and W.nnz >= n * 10
    with pytest.raises(NotImplementedError):
        compute_knn_from_nndescent(X, n_neighbors=60)
    idx._dev.close()


@pytest.mark.parametrize("n", [513, 1025, 4097])
def test_knn_awkward_sizes(oracle, n):
    """sizes one past a level / tile boundary: clamped operand rows, padded thresholds, ragged last tiles"""
    X = oracle.synth_rows(900 + n, 0, n, 512)
    assert _check_vs_oracle(oracle, X, 10) <= 2


@pytest.mark.parametrize("batched", [False, True])
def test_knn_one_past_the_row_batch(oracle, monkeypatch, batched):
    """131 073 rows: the batched path runs a second batch of ONE row; the all-rows path a ragged last tile"""
    if batched:
        monkeypatch.setenv("SSW_KNN_FORCE_BATCHED", "1")
    n = 131_073
    X = oracle.synth_rows(31, 0, n, 512)
    rows = [0, 1, 127, 128, 131_071, 131_072] + np.random.default_rng(8).integers(0, n, size=20).tolist()
    assert _check_vs_oracle(oracle, X, 10, rows=rows) <= n // 100


@pytest.mark.parametrize("n,k,dim", [(600, 10, 512), (1025, 10, 512), (3000, 10, 512), (2500, 3, 768), (2000, 31, 512),
                                     (4097, 10, 256), (5000, 15, 1024)])
def test_knn_256_tiles_every_row_vs_oracle(oracle, monkeypatch, n, k, dim):
    """the symmetric last level on 256 x 256 tiles (k_knn_gemm_filter256; taken from 32 768 rows on by default),
    forced here at sizes where every row can be held against the oracle: ragged last tiles in both directions, one
    and several super-tiles, every supported dim (4 ... 16 K-tiles per output tile)"""
    monkeypatch.setenv("SSW_KNN_TILE256_FROM", "0")
    X = oracle.synth_rows(300 + n, 0, n, dim)
    redone = _check_vs_oracle(oracle, X, k)
    assert redone <= max(2, n // 50)


def test_knn_256_tiles_clustered_duplicates(oracle, monkeypatch):
    monkeypatch.setenv("SSW_KNN_TILE256_FROM", "0")
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    X = np.concatenate([g["X"], g["X"][::-1] * np.float32(1.0)])  # every row twice: ties across tiles
    if X.shape[0] <= 512:
        X = np.concatenate([X, X, X])
    redone = _check_vs_oracle(oracle, X, 10)
    assert redone <= X.shape[0] // 5
