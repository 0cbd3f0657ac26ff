"""k-NN graph -> sparse weight matrix (host-side, one-off per index) and the KNNGraph record.

Interface of seesaw/knn_graph.py: `rbf_kernel` (:8-21), `get_weight_matrix` (:31-104),
`post_process_graph_df` (:142-168), `compute_exact_knn` (:170-191), `KNNGraph` (:246-286).
The weight matrix is built once per index and cached (the reference goes through Ray's
object store for that); the sweeps that consume it run on the GPU
(seesaw_amd.label_propagation).  The construction below is a vectorised restatement that
yields the same CSR arrays (structure and f64 values) as the reference's COO fancy-index
assignment; tests/test_graph_host.py pins it against matrices captured from the reference.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import scipy.sparse as sp


def rbf_kernel(edist: float):
    """weight = exp(-cosine_distance / edist): falls from 1 to 1/e when distance grows by edist."""
    assert edist > 0
    spread = 1.0 / edist

    def kernel(arr):
        arr = np.asarray(arr)
        assert arr.size == 0 or (arr.min() >= -0.0001 and arr.max() <= 2.0001)
        return np.exp(-(arr.astype("float64") * spread))

    return kernel


def knn_kernel(edist=2.1):
    assert edist > 0.0
    return lambda arr: (np.asarray(arr) <= edist).astype("float32")


def _symmetric_on_device(src, dst, w, n, device):
    """the symmetric CSR (sorted indices, diagonal stored as 0) built by csrc/wmatrix.hip; None when the graph has a
    vertex pair with more than two edges or a hub beyond the kernel's reach (the host path then takes over)"""
    import ctypes

    from . import _lib
    s32, d32 = np.ascontiguousarray(src, dtype=np.int32), np.ascontiguousarray(dst, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    h, nnz = ctypes.c_void_p(), ctypes.c_int64(0)
    try:
        _lib.call("ssw_wm_build_symmetric", int(device), int(n), int(s32.shape[0]), ctypes.c_void_p(s32.ctypes.data),
                  ctypes.c_void_p(d32.ctypes.data), ctypes.c_void_p(w.ctypes.data), ctypes.byref(h), ctypes.byref(nnz))
    except _lib.SeesawHipError as e:
        if e.status == _lib.SSW_ERR_UNSUPPORTED:
            return None
        raise
    try:
        indptr = np.empty(n + 1, dtype=np.int64)
        indices = np.empty(nnz.value, dtype=np.int32)
        data = np.empty(nnz.value, dtype=np.float64)
        _lib.call("ssw_wm_fetch", h, ctypes.c_void_p(indptr.ctypes.data), ctypes.c_void_p(indices.ctypes.data),
                  ctypes.c_void_p(data.ctypes.data))
    finally:
        _lib.call("ssw_wm_destroy", h)
    return sp.csr_array((data, indices, indptr), shape=(n, n))


def get_weight_matrix(df: pd.DataFrame, *, kfun, self_edges=False, normalized, laplacian=False,
                      symmetric=True, device=None):
    """edges (src_vertex, dst_vertex, distance; one self edge per vertex) -> CSR weights.

    symmetric: W_ij = (w_ij + w_ji) / (number of directed edges between i and j), i.e. the
    mean over the directions present; the diagonal is zeroed but stays stored.
    laplacian: D - W (and D^-1/2 (D - W) D^-1/2 when `normalized`).
    device: a GPU ordinal -> the symmetric matrix is assembled there (csrc/wmatrix.hip: the two COO -> CSR conversions,
    sum_duplicates and sort_indices of the host form are 7.8 s at 1.56 M vertices; the arrays come out bit-identical);
    None -> the host form below."""
    assert not self_edges
    src = df.src_vertex.values.astype(np.int64)
    dst = df.dst_vertex.values.astype(np.int64)
    n = np.unique(src).shape[0]
    assert int((src == dst).sum()) == n, "every vertex needs exactly one self edge"
    w = np.asarray(kfun(df.distance.values), dtype=np.float64)
    assert (w >= 0).all(), "edge weights must be non-negative"
    # weights that underflow to 0 keep their slot in the symmetric pattern (stored zeros),
    # exactly like the reference's adjacency-driven assignment; the asymmetric form drops them
    out = None
    if symmetric and device is not None:
        out = _symmetric_on_device(src, dst, w, n, device)
    if out is not None:
        pass  # diagonal already stored as 0, indices sorted
    elif symmetric:
        rows, cols = np.concatenate([src, dst]), np.concatenate([dst, src])
        wsum = sp.coo_array((np.concatenate([w, w]), (rows, cols)), shape=(n, n)).tocsr()
        cnt = sp.coo_array((np.ones(2 * w.shape[0]), (rows, cols)), shape=(n, n)).tocsr()
        wsum.sum_duplicates()
        cnt.sum_duplicates()
        wsum.sort_indices()
        cnt.sort_indices()
        assert np.array_equal(wsum.indices, cnt.indices) and np.array_equal(wsum.indptr, cnt.indptr)
        out = sp.csr_array((wsum.data / cnt.data, wsum.indices.copy(), wsum.indptr.copy()), shape=(n, n))
        assert np.isclose(out.diagonal(), 1.0, atol=1e-5).all(), "repeated edges mishandled"
    else:
        nz = w > 0
        out = sp.coo_array((w[nz], (src[nz], dst[nz])), shape=(n, n)).tocsr()
        out.sum_duplicates()
        out.sort_indices()
    # zero the diagonal in place (entries stay stored, as in the reference's setdiag(0.))
    rr = np.repeat(np.arange(n), np.diff(out.indptr))
    out.data[rr == out.indices] = 0.0  # (no-op for the device-built matrix)
    D = np.asarray(out.sum(axis=1)).reshape(-1)
    assert (D > 0).all(), "no zero degree nodes allowed"
    if laplacian:
        assert symmetric
        out = out.copy()
        out.data = -out.data
        out.data[rr == out.indices] = D
        if normalized:
            s = sp.dia_array((1.0 / np.sqrt(D), 0), shape=(n, n))
            out = (s @ (out @ s)).tocsr()
    out = out.tocsr()
    out.sum_duplicates()
    out.sort_indices()
    assert out.has_sorted_indices
    return out


def edge_loss(laplacian_m, labels):
    return labels @ (laplacian_m @ labels)


def get_lookup_ranges(sorted_col, nvecs):
    counts = np.bincount(np.asarray(sorted_col, dtype=np.int64), minlength=nvecs)
    return np.concatenate(([0], np.cumsum(counts)))


def post_process_graph_df(df: pd.DataFrame, nvec: int) -> pd.DataFrame:
    """normalise dtypes, clip distances at 0, rank neighbours by distance (1-based, first wins
    ties), add the rank-0 self edge to every vertex, sort by (src_vertex, dst_rank)."""
    src = df.src_vertex.values.astype("int32")
    dst = df.dst_vertex.values.astype("int32")
    dist = np.clip(df.distance.values.astype("float32"), a_min=0.0, a_max=None)
    keep = src != dst
    src, dst, dist = src[keep], dst[keep], dist[keep]
    # rank within a vertex = position after a stable sort by (src_vertex, distance).  The graph builders hand the edges
    # over in exactly that order already (rows in vertex order, neighbours by descending score): checking costs one
    # pass, the three-key sort of 17 M edges it avoids cost 3.2 s of the 5.7-s build at 1.56 M vectors
    if src.shape[0] > 1:
        same = src[1:] == src[:-1]
        in_order = bool(np.all(src[1:] >= src[:-1]) and np.all(dist[1:][same] >= dist[:-1][same]))
    else:
        in_order = True
    if not in_order:
        order = np.lexsort((np.arange(src.shape[0]), dist, src))
        src, dst, dist = src[order], dst[order], dist[order]
    n_edges = src.shape[0]
    starts = np.concatenate(([0], np.cumsum(np.bincount(src, minlength=nvec))))[:-1]
    rank = (np.arange(n_edges) - starts[src] + 1).astype("int32")
    # merged with the rank-0 self edges in (src_vertex, dst_rank) order without another sort: vertex v's block starts
    # after the edges of the vertices before it and their v self edges
    pos_edges = np.arange(n_edges, dtype=np.int64) + src.astype(np.int64) + 1
    pos_self = starts.astype(np.int64) + np.arange(nvec, dtype=np.int64)
    total = n_edges + nvec
    o_src, o_dst = np.empty(total, dtype="int32"), np.empty(total, dtype="int32")
    o_dist, o_rank = np.empty(total, dtype="float32"), np.empty(total, dtype="int32")
    ids = np.arange(nvec, dtype="int32")
    o_src[pos_edges], o_dst[pos_edges], o_dist[pos_edges], o_rank[pos_edges] = src, dst, dist, rank
    o_src[pos_self], o_dst[pos_self], o_dist[pos_self], o_rank[pos_self] = ids, ids, np.float32(0.0), 0
    return pd.DataFrame({"src_vertex": o_src, "dst_vertex": o_dst, "distance": o_dist, "dst_rank": o_rank})


def compute_exact_knn(vectors: np.ndarray, n_neighbors: int, device_index=None, device: int = 0) -> pd.DataFrame:
    """all-pairs cosine distances, k+1 nearest per row (knn_graph.py:170-191) on the GPU:
    DeviceIndex.knn (ssw_knn_build: fp16 MFMA candidate pass + exact f32 rescoring, certified
    per row).  `device_index` is the index's resident matrix when the caller has one."""
    from .device_index import DeviceIndex
    n = vectors.shape[0]
    k = min(n_neighbors + 1, n) - 1
    dev = device_index if device_index is not None else DeviceIndex.from_numpy(
        np.ascontiguousarray(vectors, dtype=np.float32), device=device)
    try:
        dst, score, _ = dev.knn(k)
    finally:
        if device_index is None:
            dev.close()
    src = np.repeat(np.arange(n, dtype=np.int32), k + 1)
    df = pd.DataFrame({"src_vertex": src, "dst_vertex": dst.reshape(-1),
                       "distance": (np.float32(1.0) - score.reshape(-1)).astype("float32")})
    return post_process_graph_df(df, nvec=n)


MAX_EXACT_K = 31  # ssw_knn_build keeps 2 (k + 1) <= 64 candidates per vertex between levels


def compute_knn_from_nndescent(vectors, *, n_neighbors, n_jobs=-1, low_memory=False, device_index=None, device: int = 0,
                               **kwargs):
    """The reference builds its production graphs with the approximate pynndescent
    (knn_graph.py:194-215, "nndescent60": 60 neighbours, later restricted to knn_k = 10 by
    KNNGraph.restrict_k).  Here the same DataFrame comes from the exact GPU builder, with pools of up to 31
    neighbours: build with at least knn_k + 1 so that restrict_k(k=knn_k) keeps `dst_rank < knn_k` as it does
    on the reference's pools (an exact graph needs no larger pool: the first knn_k - 1 neighbours do not
    depend on it).  n_jobs / low_memory are accepted for signature compatibility."""
    if n_neighbors > MAX_EXACT_K:
        raise NotImplementedError(f"exact graphs are built with n_neighbors <= {MAX_EXACT_K} (asked for "
                                  f"{n_neighbors}): a pool of knn_k + 1 gives the loops the same restricted graph")
    return compute_exact_knn(np.asarray(vectors), n_neighbors, device_index=device_index, device=device)


def build_knn_graph(index, *, name: str, n_neighbors: int = 10) -> "KNNGraph":
    """exact k-NN graph of an index's vectors, saved where lookup_weight_matrix looks for it:
    `<index.path>/knn_graph/<name>/forward.parquet` (AccessMethod.get_knng_path, interface.py:27-30)"""
    df = compute_exact_knn(index.vectors, n_neighbors, device_index=getattr(index, "_dev", None),
                           device=getattr(index, "device", 0) or 0)
    g = KNNGraph(df)
    g.save(index.get_knng_path(name=name))
    return g


def factor_neighbors(knng, idx, k_intra):
    """neighbours from other images ranked separately from neighbours inside the same image: the closest
    tile of every other image, plus up to k_intra tiles of the vertex's own image (knn_graph.py:217-243)"""
    dbidxs = idx.vector_meta.dbidx.astype("int32").values
    df = knng.knn_df
    df = df.assign(src_dbidx=dbidxs[df.src_vertex.values], dst_dbidx=dbidxs[df.dst_vertex.values])
    inter = df[df.src_dbidx != df.dst_dbidx]
    edge_ranks = inter.groupby(["src_vertex", "dst_dbidx"]).distance.rank("first").astype("int")
    inter = inter[edge_ranks <= 1]
    inter = inter.assign(dst_rank=inter.groupby(["src_vertex"]).distance.rank("first").sub(1).astype("int"))
    intra = df[df.src_dbidx == df.dst_dbidx]
    intra = intra.assign(dst_rank=intra.groupby("src_vertex").distance.rank("first").astype("int"))
    intra = intra[intra.dst_rank <= k_intra]
    return pd.concat([inter, intra], ignore_index=True)


class KNNGraph:
    def __init__(self, knn_df: pd.DataFrame, nvecs=None):
        self.knn_df = knn_df
        ks = knn_df.groupby("src_vertex").dst_rank.max()
        self._ks = ks
        self.k = ks.min()
        self.maxk = ks.median()
        self.nvecs = ks.shape[0]
        self.ind_ptr = get_lookup_ranges(knn_df.src_vertex, self.nvecs)

    def restrict_k(self, *, k):
        if k < self.maxk:
            return KNNGraph(self.knn_df[self.knn_df.dst_rank < k].reset_index(drop=True))
        assert k == self.maxk, f"can only do up to k={self.k} neighbors based on input df"
        return self

    @staticmethod
    def from_file(path):
        return KNNGraph(pd.read_parquet(f"{path}/forward.parquet"))

    def save(self, path):
        import os
        os.makedirs(path, exist_ok=True)
        self.knn_df.to_parquet(f"{path}/forward.parquet")

    def rev_lookup(self, dst_vertex) -> pd.DataFrame:
        return self.knn_df.iloc[self.ind_ptr[dst_vertex]:self.ind_ptr[dst_vertex + 1]]
