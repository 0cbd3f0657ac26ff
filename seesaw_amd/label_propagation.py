"""LabelPropagation: the reference's interface (seesaw/label_propagation.py:6-79) with the
sweeps on the GPU (ssw_labelprop_run in libseesaw_hip.so -- CSR f64 SpMV fused with the
`(+ lambda * prior) / (colsum + lambda)` normalisation, the label clamp and the device-side
`max((f'-f)^2) < epsilon` early exit)."""
from __future__ import annotations

import ctypes

import numpy as np
import scipy.sparse as sp

from . import _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def locality_order(weight_matrix, *, window: int = 32768, min_gain: float = 2.0, min_nodes: int = 1 << 19):
    """A node order (new_of_old, int32) that puts graph neighbours near each other -- reverse Cuthill-McKee over the
    symmetric pattern -- or None when it would not pay: small graphs (the iterate stays in L2 anyway), or graphs where the
    share of edges that land within `window` positions of each other does not grow `min_gain`-fold (k-NN graphs over
    unclustered vectors have no locality to find; the column-blocked sweep serves those).  One-off per graph."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    W = weight_matrix.tocsr()
    n = W.shape[0]
    if n < min_nodes:
        return None
    pattern = sp.csr_matrix((np.ones(W.nnz, dtype=np.int8), W.indices, W.indptr), shape=W.shape)
    pattern = (pattern + pattern.T).tocsr()
    old_of_new = np.asarray(reverse_cuthill_mckee(pattern, symmetric_mode=True), dtype=np.int64)
    new_of_old = np.empty(n, dtype=np.int32)
    new_of_old[old_of_new] = np.arange(n, dtype=np.int32)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(W.indptr))
    near_before = float((np.abs(rows - W.indices) < window).mean())
    near_after = float((np.abs(new_of_old[rows].astype(np.int64) - new_of_old[W.indices]) < window).mean())
    if near_after < min_gain * max(near_before, 1e-9) or near_after < 0.5:
        return None
    return new_of_old


def ordered_csr(W, weight_sum, new_of_old):
    """the CSR arrays ssw_labelprop_create_ordered takes: row r = the row of the node at position r, entries in ascending
    ORIGINAL column id (W has sorted indices), columns relabelled to positions; weight_sum in position order"""
    n = W.shape[0]
    old_of_new = np.empty(n, dtype=np.int64)
    old_of_new[new_of_old] = np.arange(n, dtype=np.int64)
    counts = np.diff(W.indptr)[old_of_new]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    # entry positions of the old rows, concatenated in new-row order
    starts = W.indptr[:-1][old_of_new].astype(np.int64)
    take = np.repeat(starts - indptr[:-1], counts) + np.arange(indptr[-1], dtype=np.int64)
    indices = np.ascontiguousarray(new_of_old[W.indices[take]], dtype=np.int32)
    data = np.ascontiguousarray(W.data[take], dtype=np.float64)
    wsum = np.ascontiguousarray(np.asarray(weight_sum, dtype=np.float64)[old_of_new])
    return indptr, indices, data, wsum


class LabelPropagation:
    collect_run_info = False  # round(): also read ssw_labelprop_last_run_info (bench.py's phase timers; one more C call)

    def __init__(self, weight_matrix, *, reg_lambda: float, max_iter: int, epsilon=1e-5, verbose=0, device: int = 0,
                 node_order=None):
        """node_order: optional permutation new_of_old (locality_order(weight_matrix)) -- the graph is then stored on the
        device in that order (rows' entries still in ascending original column id, so every result stays bit-identical)
        and the sweep's gathers of the iterate hit cache on clustered data; all ids at this interface stay original."""
        assert reg_lambda >= 0
        W = weight_matrix if sp.issparse(weight_matrix) else sp.csr_array(weight_matrix)
        W = W.tocsr()
        assert W.has_sorted_indices
        self.weight_matrix = W
        self.n = W.shape[0]
        self.epsilon = epsilon
        self.verbose = verbose
        self.reg_lambda = float(reg_lambda)
        self.max_iter = int(max_iter)
        self.device = int(device)
        self.reg_values = None
        self.weight_sum = np.asarray(W.sum(0)).reshape(-1).astype(np.float64)  # column sums, as the reference
        self.last_sweeps = 0
        self.last_converged = False
        self._prior_installed = False
        self._h = ctypes.c_void_p()
        self.node_order = None
        if node_order is not None:
            new_of_old = np.ascontiguousarray(node_order, dtype=np.int32)
            assert new_of_old.shape == (self.n,)
            indptr, indices, data, wsum = ordered_csr(W, self.weight_sum, new_of_old)
            _lib.call("ssw_labelprop_create_ordered", int(device), self.n, _p(indptr), _p(indices), _p(data), _p(wsum),
                      _p(new_of_old), ctypes.byref(self._h))
            self.node_order = new_of_old
            return
        indptr = np.ascontiguousarray(W.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(W.indices, dtype=np.int32)
        data = np.ascontiguousarray(W.data, dtype=np.float64)
        _lib.call("ssw_labelprop_create", int(device), self.n, _p(indptr), _p(indices), _p(data),
                  _p(self.weight_sum), ctypes.byref(self._h))

    def close(self):
        if self._h:
            _lib.load().ssw_labelprop_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device-resident chaining (the ranking loop: same prior every call, result feeds a top-k) ----
    def set_prior(self, reg_values):
        """install reg_values on the device once; fit_resident then uses it as prior AND start iterate"""
        assert reg_values.shape[0] == self.n
        self.reg_values = reg_values
        prior = np.ascontiguousarray(reg_values, dtype=np.float64)  # (a named reference: _p keeps only the address)
        _lib.call("ssw_labelprop_set_prior", self._h, _p(prior))
        self._prior_installed = True

    def fit_resident(self, *, label_ids, label_values):
        """fit_transform(label_ids, label_values, reg_values=prior, start_value=prior) with the installed
        prior; the result stays on the device (fetch() / scores_to_index())."""
        ids = np.ascontiguousarray(np.asarray(label_ids).reshape(-1), dtype=np.int64)
        vals = np.ascontiguousarray(np.asarray(label_values).reshape(-1), dtype=np.float64)
        assert ids.shape == vals.shape
        sweeps, conv = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.call("ssw_labelprop_run_resident", self._h, _p(ids), _p(vals), ids.shape[0], self.reg_lambda,
                  float(self.epsilon), self.max_iter, ctypes.byref(sweeps), ctypes.byref(conv))
        self.last_sweeps, self.last_converged = sweeps.value, bool(conv.value)
        self._read_run_info()
        if self.last_converged:
            if self.verbose > 0:
                print(f"prop. converged after {self.last_sweeps} iterations")
        else:
            print(f"warning: did not converge after {self.last_sweeps} iterations")

    def round(self, device_index, *, propagate: bool, label_ids, label_values, mask_labeled: bool, excluded, k: int):
        """One feedback round in one C-ABI call (ssw_labelprop_round): propagate the labels over the installed prior
        (or, propagate=False, make the prior the result with `label_ids` marked), write the f32 scores into the index's
        score buffer and select the top k distinct non-excluded images there.  Same results as fit_resident /
        prior_as_result + scores_to_index + DeviceIndex.topk(None, k, excluded); one host wait when the propagation is
        an incremental update.  -> (image positions, scores f32, best rows)"""
        ids = np.ascontiguousarray(label_ids, dtype=np.int64).reshape(-1)
        vals = np.ascontiguousarray(label_values, dtype=np.float64).reshape(-1)
        assert ids.shape == vals.shape
        ex = None if excluded is None else np.ascontiguousarray(excluded, dtype=np.int64)
        n_ex = 0 if ex is None else ex.shape[0]
        k = int(k)
        st = self.__dict__.get("_round_out")
        if st is None or st[0].shape[0] < k:  # outputs and their addresses, made once (ndarray.ctypes costs ~1.5 us an access)
            outs = (np.empty(k, dtype=np.int64), np.empty(k, dtype=np.float32), np.empty(k, dtype=np.int64))
            cnt, sweeps, conv = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
            st = self._round_out = outs + (cnt, sweeps, conv, tuple(ctypes.c_void_p(a.ctypes.data) for a in outs),
                                           (ctypes.byref(cnt), ctypes.byref(sweeps), ctypes.byref(conv)))
        imgs, scs, rows, cnt, sweeps, conv, out_ptrs, refs = st
        addr = lambda a: ctypes.c_void_p(a.__array_interface__["data"][0])  # noqa: E731
        _lib.call("ssw_labelprop_round", self._h, device_index._h, int(bool(propagate)), addr(ids), addr(vals), ids.shape[0],
                  self.reg_lambda, float(self.epsilon), self.max_iter, int(bool(mask_labeled)), addr(ex) if n_ex else None, n_ex, k,
                  out_ptrs[0], out_ptrs[1], out_ptrs[2], refs[0], refs[1], refs[2])
        if propagate:
            self.last_sweeps, self.last_converged = sweeps.value, bool(conv.value)
            if not self.last_converged:
                print(f"warning: did not converge after {self.last_sweeps} iterations")
            elif self.verbose > 0:
                print(f"prop. converged after {self.last_sweeps} iterations")
        if self.collect_run_info:
            self._read_run_info()
        c = cnt.value
        return imgs[:c].copy(), scs[:c].copy(), rows[:c].copy()

    def _read_run_info(self):
        """what the propagation just did on the device (ssw_labelprop_last_run_info): consecutive fit_resident calls
        are incremental -- only rows within k hops of a changed label are recomputed for sweep k, bit-identical results"""
        info = np.zeros(8, dtype=np.int64)
        _lib.call("ssw_labelprop_last_run_info", self._h, _p(info))
        self.last_mode = int(info[0])              # 0 full sweeps, 1 incremental update, 2 incremental pass continued by full sweeps
        self.last_incremental = self.last_mode == 1
        self.last_launches, self.last_host_syncs, self.last_rows_recomputed = int(info[2]), int(info[3]), int(info[4])
        self.last_frontier_us, self.last_device_wait_us = info[6] / 1e3, info[7] / 1e3

    def prior_as_result(self, label_ids):
        """the installed prior becomes the resident result, `label_ids` are marked labelled (nothing is propagated)"""
        ids = np.ascontiguousarray(np.asarray(label_ids).reshape(-1), dtype=np.int64)
        _lib.call("ssw_labelprop_prior_as_result", self._h, _p(ids), ids.shape[0])

    def fetch(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        _lib.call("ssw_labelprop_fetch", self._h, _p(out))
        return out

    def gather(self, rows) -> np.ndarray:
        """result[rows] of the last propagation: 8 bytes per asked node cross PCIe, not the whole iterate"""
        rows = np.ascontiguousarray(np.asarray(rows).reshape(-1), dtype=np.int64)
        out = np.empty(rows.shape[0], dtype=np.float64)
        _lib.call("ssw_labelprop_gather", self._h, _p(rows), rows.shape[0], _p(out))
        return out

    def device_scores_ptr(self) -> int:
        """device address of the f64 result of the last propagation"""
        out = ctypes.c_void_p()
        _lib.call("ssw_labelprop_device_scores", self._h, ctypes.byref(out))
        return int(out.value)

    def scores_to_index(self, device_index, mask_labeled: bool = True):
        """last result -> the index's f32 score buffer (labelled nodes at -inf), ready for topk(None, ...)"""
        _lib.call("ssw_labelprop_scores_to_index", self._h, device_index._h, int(bool(mask_labeled)))

    def fit_transform(self, *, label_ids, label_values, reg_values=None, start_value=None):
        self._prior_installed = False  # this call overwrites the device copy of the prior
        if reg_values is not None:
            assert reg_values.shape[0] == self.n
            self.reg_values = reg_values
            prior = np.ascontiguousarray(reg_values, dtype=np.float64)
        else:
            assert self.reg_lambda == 0
            self.reg_values = np.zeros(self.n)
            prior = None
        if start_value is not None:
            start = np.array(start_value, dtype=np.float64)
        elif reg_values is not None:
            start = np.array(reg_values, dtype=np.float64)
        else:
            start = np.zeros(self.n)
        ids = np.ascontiguousarray(np.asarray(label_ids).reshape(-1), dtype=np.int64)
        vals = np.ascontiguousarray(np.asarray(label_values).reshape(-1), dtype=np.float64)
        assert ids.shape == vals.shape
        out = np.empty(self.n, dtype=np.float64)
        sweeps, conv = ctypes.c_int32(0), ctypes.c_int32(0)
        start = np.ascontiguousarray(start)
        _lib.call("ssw_labelprop_run", self._h, _p(prior), _p(start), _p(ids), _p(vals),
                  ids.shape[0], self.reg_lambda, float(self.epsilon), self.max_iter, _p(out),
                  ctypes.byref(sweeps), ctypes.byref(conv))
        self.last_sweeps, self.last_converged = sweeps.value, bool(conv.value)
        self._read_run_info()
        if self.last_converged:
            if self.verbose > 0:
                print(f"prop. converged after {self.last_sweeps} iterations")
        else:
            print(f"warning: did not converge after {self.last_sweeps} iterations")
        return out
