"""KnnProp2: rank by label propagation over the k-NN graph (seesaw/loops/graph_based.py:18-121),
plus the weight-matrix / X'LX lookup shared with MultiReg and PseudoLR."""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sp
from pydantic import BaseModel

from ..bitmap import BitMap
from ..indices.multiscale.multiscale_index import rescore_candidates
from ..knn_graph import KNNGraph, get_weight_matrix, rbf_kernel
from ..research.knn_methods import LabelPropagationRanker2
from .loop_base import LoopBase

_CACHE = {}  # process-local stand-in for the reference's Ray-backed `_cache_closure`


class WeightMatrixOptions(BaseModel):
    knn_path: str
    knn_k: int
    edist: float
    self_edges: bool
    normalized_weights: bool
    symmetric: bool
    xlx_matrix: bool = False


def compute_xlx(laplacian, X_vectors: np.ndarray, device_index=None) -> np.ndarray:
    """X' (L / trace L) X -- the database-alignment regulariser of MultiReg
    (graph_based.py:45-49).  One-off per index, cached.  Runs on the GPU (ssw_xlx: f64 CSR
    SpMM + f64 MFMA product) against the matrix already resident for the scan; a temporary
    device copy of X_vectors is made when the caller has no DeviceIndex."""
    import ctypes

    from .. import _lib
    from ..device_index import DeviceIndex
    from ..label_propagation import LabelPropagation
    L = sp.csr_array(laplacian / laplacian.diagonal().sum())
    L.sort_indices()
    dev = device_index if device_index is not None else DeviceIndex.from_numpy(np.ascontiguousarray(X_vectors, dtype=np.float32))
    lap = LabelPropagation(L, reg_lambda=0.0, max_iter=0, device=dev.device)
    out = np.empty((dev.dim, dev.dim), dtype=np.float64)
    try:
        _lib.call("ssw_xlx", dev._h, lap._h, ctypes.c_void_p(out.ctypes.data))
    finally:
        lap.close()
        if device_index is None:
            dev.close()
    return out


def lookup_weight_matrix(opts: WeightMatrixOptions, *, use_cache: bool, X_vectors=None, knng: KNNGraph = None,
                         device_index=None, device=None):
    key = opts.model_dump_json()
    if opts.xlx_matrix:
        assert opts.symmetric and not opts.self_edges
    if use_cache and key in _CACHE:
        return _CACHE[key]
    print(f"init weight matrix {opts=}")
    graph = (knng if knng is not None else KNNGraph.from_file(opts.knn_path)).restrict_k(k=opts.knn_k)
    wm = get_weight_matrix(graph.knn_df, kfun=rbf_kernel(opts.edist), self_edges=opts.self_edges,
                           normalized=opts.normalized_weights, symmetric=opts.symmetric, laplacian=opts.xlx_matrix,
                           device=device)  # the symmetric assembly runs on the index's GPU (csrc/wmatrix.hip)
    if opts.xlx_matrix:
        assert X_vectors is not None
        wm = compute_xlx(wm, X_vectors, device_index=device_index)
    if use_cache:
        _CACHE[key] = wm
    return wm


def get_weight_matrix_from_index(idx, weight_matrix_options, xlx_matrix=False):
    opts = WeightMatrixOptions(**weight_matrix_options)
    opts.xlx_matrix = xlx_matrix
    knng = getattr(idx, "knng", {}).get(opts.knn_path) if isinstance(getattr(idx, "knng", None), dict) else None
    if knng is None:
        opts.knn_path = os.path.normpath(os.path.abspath(os.path.realpath(idx.get_knng_path(name=opts.knn_path))))
    else:  # in-memory graph attached to the index (synthetic datasets): key the cache by identity
        opts.knn_path = f"mem:{id(idx)}:{opts.knn_path}"
    use_cache = opts.knn_path.find("subset") == -1
    return lookup_weight_matrix(opts, use_cache=use_cache, X_vectors=idx.vectors, knng=knng,
                                device_index=getattr(idx, "_dev", None) if xlx_matrix else None,
                                device=getattr(idx, "device", None) if (getattr(idx, "_dev", None) is not None or
                                                                        getattr(idx, "_shard", None) is not None) else None)


_ORDER = {}  # id(weight matrix) -> (the matrix, its locality order or None): one-off per graph, like the matrix itself


def _locality_order_of(W):
    """label_propagation.locality_order(W), remembered next to the cached weight matrix (large clustered graphs are
    stored on the device in a neighbour-preserving node order; everything else returns None)"""
    if os.environ.get("SSW_LP_NO_REORDER"):
        return None
    hit = _ORDER.get(id(W))
    if hit is None or hit[0] is not W:
        from ..label_propagation import locality_order
        hit = _ORDER[id(W)] = (W, locality_order(W))
    return hit[1]


def get_label_prop(q, label_prop_params):
    W = get_weight_matrix_from_index(q.index, label_prop_params["matrix_options"])
    params = {k: v for k, v in label_prop_params.items() if k != "matrix_options"}
    return LabelPropagationRanker2(weight_matrix=W, device=getattr(q.index, "device", 0), node_order=_locality_order_of(W),
                                   **params)


class KnnProp2(LoopBase):
    def __init__(self, gdm, q, params, knn_model):
        super().__init__(gdm, q, params)
        self.state.knn_model = knn_model

    @staticmethod
    def from_params(gdm, q, p):
        return KnnProp2(gdm, q, p, get_label_prop(q, p.interactive_options))

    def set_text_vec(self, tvec):
        super().set_text_vec(tvec)
        self._pending = None  # a shortlist selected under the previous query's prior is void
        self.state.knn_model.set_base_scores(self.q.index.score(tvec))

    def next_batch(self):
        """next images by propagated score: unlabelled vectors only, distinct non-returned
        images, then the usual per-image aggregation (graph_based.py:88-109)."""
        model, p, q = self.state.knn_model, self.params, self.q
        resident = getattr(model, "scores_on_device", lambda: False)() and \
            callable(getattr(q.index, "topk_from_device_scores", None))
        avg_on_device = resident and p.agg_method != "plain_score" and getattr(q.index, "_has_tile_meta", False) \
            and getattr(p, "aug_weight", None) in (None, "level_max") and hasattr(q.index, "rescore_avg_from_device_scores")
        on_device = resident and (p.agg_method == "plain_score" or avg_on_device)
        pending, self._pending = getattr(self, "_pending", None), None
        if on_device and pending is not None and pending[1] == len(q.returned) and pending[2] == p.shortlist_size \
                and pending[3] == getattr(model, "_labels_stamp", 0):
            cand = pending[0]  # refine() propagated AND selected in one device call (ssw_labelprop_round)
            scores = None
        elif on_device:  # propagated scores go from the graph handle to the index's score buffer on the GPU
            cand = q.index.topk_from_device_scores(lambda dev: model.lp.scores_to_index(dev, mask_labeled=True),
                                                   topk_dbidx=p.shortlist_size, exclude_dbidx=q.returned)
            scores = None
        else:
            scores = model.current_scores()
            cand = q.index.topk_from_scores(scores, topk_dbidx=p.shortlist_size, exclude_dbidx=q.returned,
                                            skip_rows=model.is_labeled > 0)
        if p.agg_method == "plain_score":  # best tile per image came back with the selection
            ans = q.index._activations_from_best(cand, p.batch_size)
        elif avg_on_device:  # the averaging aggregation over the UNMASKED f64 scores, where they are
            ans = q.index.rescore_avg_from_device_scores(cand, p.batch_size, p.aug_larger, model.lp.device_scores_ptr())
        else:
            rows = q.index._candidate_rows(cand.attrs["positions"])
            fullmeta = q.index.vector_meta.iloc[rows].assign(score=np.asarray(scores)[rows])
            ans = rescore_candidates(fullmeta, topk=p.batch_size, **p.dict())
        q.returned.update(np.asarray(ans["dbidxs"], dtype=np.int64))
        return ans

    def refine(self, change=None):
        pos, neg = self.q.getXy(get_positions=True)
        idxs = np.concatenate([pos, neg])
        labels = np.concatenate([np.ones_like(pos), np.zeros_like(neg)])
        model, q, p = self.state.knn_model, self.q, self.params
        self._pending = None
        if not os.environ.get("SSW_NO_FUSED_ROUND") and callable(getattr(q.index, "topk_after_update", None)) \
                and getattr(q.index, "_dev", None) is not None and getattr(model, "can_fuse_round", lambda: False)():
            # the update and the selection the next next_batch() will ask for, in ONE device call and one wait: nothing
            # next_batch's selection depends on (labels, returned images, shortlist size) changes in between
            cand = q.index.topk_after_update(model, idxs, labels, topk_dbidx=p.shortlist_size, exclude_dbidx=q.returned)
            self._pending = (cand, len(q.returned), p.shortlist_size, getattr(model, "_labels_stamp", 0))
        else:
            model.update(idxs, labels)
