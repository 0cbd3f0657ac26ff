"""GPU: the symmetric k-NN weight matrix assembled on the device (csrc/wmatrix.hip, get_weight_matrix(device=...))
against the matrices captured from the reference's get_weight_matrix (tests/golden/labelprop.npz) and against the host
form on larger graphs with hub vertices, asymmetric neighbourhoods and underflowing weights -- CSR arrays bit for bit."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _same(A, B):
    assert A.shape == B.shape
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    assert np.array_equal(A.data.view(np.uint64), B.data.view(np.uint64))


def test_reference_golden_matrices_bit_for_bit():
    from seesaw_amd.knn_graph import KNNGraph, get_weight_matrix, rbf_kernel
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    df = pd.DataFrame({"src_vertex": g["src"], "dst_vertex": g["dst"], "distance": g["dist"], "dst_rank": g["rank"]})
    kg = KNNGraph(df).restrict_k(k=int(g["k"]))
    for name, ed in [("e05", 0.05), ("e10", 0.1)]:
        W = get_weight_matrix(kg.knn_df, kfun=rbf_kernel(ed), self_edges=False, normalized=False, symmetric=True, device=0)
        assert np.array_equal(W.indptr, g[f"{name}_indptr"]) and np.array_equal(W.indices, g[f"{name}_indices"])
        assert np.array_equal(W.data, g[f"{name}_data"])
    L = get_weight_matrix(kg.knn_df, kfun=rbf_kernel(0.05), self_edges=False, normalized=False, symmetric=True,
                          laplacian=True, device=0)
    assert np.array_equal(L.data, g["lap_data"]) and np.array_equal(L.indices, g["lap_indices"])


def _random_graph(n, k, seed, hub=None):
    rng = np.random.default_rng(seed)
    src = np.repeat(np.arange(n, dtype=np.int32), k + 1)
    dst = np.empty((n, k + 1), dtype=np.int64)
    for i in range(n):  # k distinct neighbours != i (no repeated edges), self edge first
        nb = rng.choice(n - 1, size=k, replace=False)
        nb[nb >= i] += 1
        dst[i, 0], dst[i, 1:] = i, nb
    if hub is not None:  # everybody's first neighbour is the hub: in-degree n - 1 (the workgroup-per-row path)
        for i in range(n):
            if i != hub and hub not in dst[i, 1:]:
                dst[i, 1] = hub
    dist = np.sort(rng.random((n, k + 1)).astype(np.float32) * 0.8, axis=1)
    dist[:, 0] = 0
    rank = np.tile(np.arange(k + 1, dtype=np.int32), n)
    return pd.DataFrame({"src_vertex": src, "dst_vertex": dst.reshape(-1).astype(np.int32), "distance": dist.reshape(-1),
                         "dst_rank": rank})


@pytest.mark.parametrize("n,k,hub,edist", [(5000, 10, None, 0.05), (3000, 10, 7, 0.05), (4000, 24, None, 0.1),
                                           (2000, 10, None, 0.001)])
def test_device_equals_host_form(n, k, hub, edist):
    from seesaw_amd.knn_graph import get_weight_matrix, rbf_kernel
    df = _random_graph(n, k, seed=n + k, hub=hub)
    kw = dict(kfun=rbf_kernel(edist), self_edges=False, normalized=False, symmetric=True)
    try:
        host = get_weight_matrix(df, **kw)
    except AssertionError:  # (edist 0.001: weights underflow to zero-degree nodes, the reference's own assert)
        with pytest.raises(AssertionError):
            get_weight_matrix(df, device=0, **kw)
        return
    dev = get_weight_matrix(df, device=0, **kw)
    _same(host, dev)
    if hub is not None:
        assert np.diff(dev.indptr).max() >= n - 1


def test_repeated_edges_fall_back_to_the_host_form():
    """a vertex pair with more than two edges: the device path declines (order-dependent f64 sum) and
    get_weight_matrix falls back to the host form, whose own assertion about repeated edges then speaks"""
    from seesaw_amd.knn_graph import _symmetric_on_device
    src = np.array([0, 0, 0, 1, 1, 2, 2], dtype=np.int64)
    dst = np.array([0, 1, 1, 1, 0, 2, 0], dtype=np.int64)
    w = np.array([1.0, 0.5, 0.25, 1.0, 0.125, 1.0, 0.3])
    assert _symmetric_on_device(src, dst, w, 3, 0) is None
