"""LKNNModel: the L-KNN probability model of the active-search loops (seesaw/loops/LKNN_model.py:76-281).

p_i = (gamma_i + sum of the labels of i's labelled neighbours) / (1 + number of labelled neighbours).  The state is
three host arrays (numerators, denominators, gamma) over the nodes and the CSR neighbour lists; conditioning on a
label touches one row of neighbours.  What is expensive in the reference -- the look-ahead value of every node,
an N x (K + 2D) argsort per planning step -- runs on the GPU (`top_sum`, ssw_lknn_top_sum).  The descending order of the
scores is sorted in full only where the scores are new as a whole (from_dataset, with_gamma: once per text query);
after an answer (condition_) the handful of changed nodes is merged back into it."""
from __future__ import annotations

import ctypes
import math
from typing import Tuple

import numpy as np
import scipy.sparse as sp

from .. import _lib
from ..bitmap import FrozenBitMap
from ..research.active_search.common import Dataset, ProbabilityModel


def initial_gamma_array(gamma, shape):
    """the prior, jittered by 1e-6 so that no two nodes tie (LKNN_model.py:70-73)"""
    return np.random.default_rng(seed=0).normal(loc=gamma, scale=1e-6, size=shape)


def _reinsert_sorted(desc_idx, desc_score, changed, score):
    """the descending order after the scores of a handful of nodes changed: those nodes are taken out of the order and
    merged back at their new places -- O(N) copies instead of the reference's full `np.argsort(-score)` per answer
    (seesaw/loops/LKNN_model.py:185: ~0.1 s at 1.56 M nodes).  Equal to that argsort whenever the scores are distinct (the
    prior is jittered so that they are, LKNN_model.py:70-73); among exactly equal scores a node that moved goes before
    the ones that stayed, where introsort's order is unspecified."""
    changed = np.unique(np.asarray(changed, dtype=np.int64))
    if changed.shape[0] == 0:
        return desc_idx, desc_score
    stay = np.ones(score.shape[0], dtype=bool)
    stay[changed] = False
    keep = stay[desc_idx]
    old_idx, old_score = desc_idx[keep], desc_score[keep]
    new_score = score[changed]
    order = np.argsort(-new_score, kind="stable")
    changed, new_score = changed[order], new_score[order]
    at = np.searchsorted(-old_score, -new_score, side="left")  # before the stayers of equal score
    return np.insert(old_idx, at, changed), np.insert(old_score, at, new_score)


class LKNNModel(ProbabilityModel):
    def __init__(self, dataset: Dataset, gamma, matrix: sp.csr_array, numerators, denominators, score, desc_idx, desc_score,
                 desc_changed_idx, desc_changed_score, device: int = 0, _gpu=None):
        super().__init__(dataset)
        self.matrix, self.numerators, self.denominators, self.score, self.gamma = matrix, numerators, denominators, score, gamma
        assert gamma.shape == self.numerators.shape
        assert ((0 < gamma) & (gamma < 1)).all(), "this could fail by chance, decrase var. or fix properly by applying sigmoid"
        self.desc_idx, self.desc_score = desc_idx, desc_score
        self.desc_changed_idx, self.desc_changed_score = desc_changed_idx, desc_changed_score  # filled by condition()
        self.device = device
        self._gpu = _gpu if _gpu is not None else {}  # shared by the models derived from one another: one device handle
        self._init_sets()

    def _init_sets(self):
        self.changed_idx_set = FrozenBitMap(self.desc_changed_idx) if self.desc_changed_idx is not None else FrozenBitMap()
        self.ignore_set = self.dataset.seen_indices.union(self.changed_idx_set)

    @staticmethod
    def from_dataset(dataset: Dataset, weight_matrix: sp.csr_array, gamma: np.ndarray, device: int = 0):
        assert weight_matrix.format == "csr"
        assert len(dataset.idx2label) == 0, "not implemented other case"
        sz = weight_matrix.shape[0]
        assert gamma.shape == (sz,)
        init_scores = (np.zeros(sz) + gamma) / (np.zeros(sz) + 1)
        order = np.argsort(-init_scores)
        return LKNNModel(dataset, gamma=gamma, matrix=weight_matrix, numerators=np.zeros(sz), denominators=np.zeros(sz),
                         score=init_scores, desc_idx=order, desc_score=init_scores[order], desc_changed_idx=None,
                         desc_changed_score=None, device=device)

    def _derive(self, **kw):
        base = dict(dataset=self.dataset, gamma=self.gamma, matrix=self.matrix, numerators=self.numerators,
                    denominators=self.denominators, score=self.score, desc_idx=self.desc_idx, desc_score=self.desc_score,
                    desc_changed_idx=self.desc_changed_idx, desc_changed_score=self.desc_changed_score, device=self.device,
                    _gpu=self._gpu)
        base.update(kw)
        return LKNNModel(**base)

    # ---- conditioning -------------------------------------------------------------------------
    def _condition_shared(self, idx, y, ret_num_denom=False):
        start, end = self.matrix.indptr[idx:idx + 2]
        neighbors = self.matrix.indices[start:end]
        assert self.dataset.idx2label.get(idx, None) is None, "no benchmark scenario should reach this"
        num, den, gam = self.numerators[neighbors], self.denominators[neighbors], self.gamma[neighbors]
        change = (num + y + gam) / (den + 1 + 1)
        order = np.argsort(-change)
        nd = (num + y, den + 1) if ret_num_denom else (None, None)
        return self.dataset.with_label(idx, y), neighbors, neighbors[order], change[order], change, nd[0], nd[1]

    def condition(self, idx, y) -> "LKNNModel":
        """a NEW model with idx labelled y: the base arrays are shared, the changed neighbours ride along sorted"""
        assert self.desc_changed_idx is None
        ds, _, chg_idx, chg_score, _, _, _ = self._condition_shared(idx, y)
        return self._derive(dataset=ds, desc_changed_idx=chg_idx, desc_changed_score=chg_score)

    def with_gamma(self, new_gamma: np.ndarray) -> "LKNNModel":
        assert self.desc_changed_idx is None, "check this is correct"
        scores = (self.numerators + new_gamma) / (self.denominators + 1)
        order = np.argsort(-scores)
        return self._derive(gamma=new_gamma, score=scores, desc_idx=order, desc_score=scores[order])

    def condition_(self, idx, y):
        """in place (after the user's answer): neighbours' counts and scores, then the descending order"""
        assert self.desc_changed_idx is None
        ds, neighbors, _, _, change, num, den = self._condition_shared(idx, y, ret_num_denom=True)
        self.dataset = ds
        self.numerators[neighbors], self.denominators[neighbors], self.score[neighbors] = num, den, change
        self.desc_idx, self.desc_score = _reinsert_sorted(self.desc_idx, self.desc_score, neighbors, self.score)
        self._init_sets()

    # ---- queries ------------------------------------------------------------------------------
    def predict_proba(self, idxs) -> np.ndarray:
        assert self.desc_changed_idx is None, "is this ever called after first round"
        return self.score[np.asarray(idxs, dtype=np.int64)]

    def _iter_desc_scores(self):
        for idx, score in zip(self.desc_idx, self.desc_score):
            if idx not in self.ignore_set:
                yield idx, score

    def _iter_changed_scores(self):
        for idx, score in zip(self.desc_changed_idx, self.desc_changed_score):
            if idx not in self.dataset.seen_indices:
                yield idx, score

    def iter_desc(self):
        """both streams merged in descending order, without the nodes already seen"""
        if self.desc_changed_idx is None:
            yield from self._iter_desc_scores()
            return
        it1, it2 = self._iter_desc_scores(), self._iter_changed_scores()
        end = (-1, -math.inf)
        (i1, s1), (i2, s2) = next(it1, end), next(it2, end)
        while i1 > -1 or i2 > -1:
            if s1 >= s2:
                yield i1, s1
                i1, s1 = next(it1, end)
            else:
                yield i2, s2
                i2, s2 = next(it2, end)

    def top_k_remaining(self, top_k: int) -> Tuple[np.ndarray, np.ndarray]:
        idxs, vals = [], []
        for i, (idx, val) in enumerate(self.iter_desc()):
            if i >= top_k:
                break
            idxs.append(idx)
            vals.append(val)
        return np.array(idxs), np.array(vals)

    def probability_bound(self, n):
        idxs = np.asarray(self.dataset.remaining_indices(), dtype=np.int64)
        return np.max((self.gamma[idxs] + n + self.numerators[idxs]) / (1 + n + self.denominators[idxs]))

    # ---- the look-ahead on the GPU --------------------------------------------------------------
    def regular_degree(self) -> int:
        deltas = np.diff(self.matrix.indptr)
        assert (deltas == deltas[0]).all(), "the vectorised look-ahead needs the same number of neighbours for every node"
        return int(deltas[0])

    # what ssw_lknn_top_sum keeps per thread (csrc/lknn.hip): K picks, D neighbours, a K + D list with a 192-bit mask
    KERNEL_MAX_K, KERNEL_MAX_D, KERNEL_MAX_LIST = 128, 32, 192

    @staticmethod
    def _top_sum_host(numer, denom, scores, nbr_sorted, K, top, block=4096):
        """the look-ahead values on the host, for horizons / degrees beyond the kernel's registers (reward_horizon > 129,
        more than 32 neighbours): same quantities, rows in blocks.  For node i the pool is the global list `top` without i
        and without i's neighbours, plus the neighbours at their conditioned scores; value = sum of its K best."""
        N, D = nbr_sorted.shape
        t_sorted_pos = np.argsort(top, kind="stable")
        t_ids = top[t_sorted_pos].astype(np.int64)
        t_scores = scores[t_ids]
        given0 = numer / (denom + 1)
        given1 = (numer + 1) / (denom + 1)
        out = np.empty(N, dtype=np.float64)
        for a in range(0, N, block):
            b = min(N, a + block)
            rows = np.arange(a, b, dtype=np.int64)[:, None]
            nb = nbr_sorted[a:b].astype(np.int64)
            base = np.broadcast_to(t_scores, (b - a, t_ids.shape[0])).copy()
            base[t_ids[None, :] == rows] = -np.inf                      # the node itself leaves the list
            pos = np.searchsorted(t_ids, nb)                            # neighbours that sit in the list are replaced
            hit = np.take(np.append(t_ids, N), pos) == nb
            rr, cc = np.nonzero(hit)
            base[rr, pos[rr, cc]] = -np.inf
            vals = []
            for cond in (given1, given0):
                ns = np.take(cond, nb)
                ns[nb == rows] = -np.inf
                pool = np.concatenate([base, ns], axis=1)
                best = -np.sort(-pool, axis=1)[:, :K]                   # descending: the order the reference adds them in
                assert (best > -np.inf).all(), "fewer than K candidates left for some node"
                vals.append(best.sum(axis=1))
            s = scores[a:b]
            out[a:b] = s * (1 + vals[0]) + (1 - s) * vals[1]
        return out

    def top_sum(self, K: int, return_values: bool = False):
        """_top_sum (efficient_nonmyopic_search.py:94-169) for the current state: value of labelling every node next,
        K further picks into the future.  -> (best index, best value[, all values])"""
        D = self.regular_degree()
        N = self.matrix.shape[0]
        h = self._gpu.get("handle")
        if h is None:
            nbr = np.ascontiguousarray(np.sort(self.matrix.indices.reshape(-1, D)), dtype=np.int32)
            h = ctypes.c_void_p()
            _lib.call("ssw_lknn_create", int(self.device), N, D, ctypes.c_void_p(nbr.ctypes.data), ctypes.byref(h))
            self._gpu["handle"] = h
            self._gpu["owner"] = _Handle(h)
        numer = self.numerators + self.gamma
        denom = self.denominators + 1
        seen = np.asarray(self.dataset.seen_indices, dtype=np.int64)
        numer[seen] = -math.inf  # will rank lowest
        assert (numer <= denom).all()
        scores = numer / denom
        L = K + D
        assert N - seen.shape[0] >= L, f"{N - seen.shape[0]} unseen nodes cannot fill a look-ahead list of K + D = {L}"
        part = np.argpartition(-scores, L - 1)[:L]          # the K + D best, then ordered: O(N) instead of a full sort
        top = part[np.argsort(-scores[part], kind="stable")].astype(np.int32)
        assert np.isfinite(scores[top]).all()               # (the reference's `top_k_scores > -inf`)
        if K > self.KERNEL_MAX_K or D > self.KERNEL_MAX_D or L > self.KERNEL_MAX_LIST:
            with np.errstate(invalid="ignore"):
                values = self._top_sum_host(numer, denom, scores, np.sort(self.matrix.indices.reshape(-1, D)), K, top)
            best = int(np.nanargmax(values))
            return (best, float(values[best]), values) if return_values else (best, float(values[best]))
        numer, denom = np.ascontiguousarray(numer), np.ascontiguousarray(denom)
        values = np.empty(N, dtype=np.float64) if return_values else None
        best_i, best_v = ctypes.c_int64(-1), ctypes.c_double(0.0)
        _lib.call("ssw_lknn_top_sum", h, ctypes.c_void_p(numer.ctypes.data), ctypes.c_void_p(denom.ctypes.data),
                  ctypes.c_void_p(top.ctypes.data), int(K), None if values is None else ctypes.c_void_p(values.ctypes.data),
                  ctypes.byref(best_i), ctypes.byref(best_v))
        return (int(best_i.value), float(best_v.value), values) if return_values else (int(best_i.value), float(best_v.value))


class _Handle:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            _lib.load().ssw_lknn_destroy(self.h)
        except Exception:
            pass
