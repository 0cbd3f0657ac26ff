"""Rankers driven by the k-NN graph (interface of seesaw/research/knn_methods.py:8-199).

LabelPropagationRanker2: calibrates the scan scores into a prior, propagates user labels
over the weight matrix (GPU sweeps through seesaw_amd.label_propagation) and ranks the
unlabelled vectors by the propagated score.
"""
from __future__ import annotations

import numpy as np
from scipy.special import expit as sigmoid

from ..bitmap import FrozenBitMap
from ..knn_graph import KNNGraph  # noqa: F401  (re-exported like the reference)
from ..label_propagation import LabelPropagation


def normalize_scores(scores, epsilon):
    """affine map of the scores onto [epsilon, 1 - epsilon]; constant input -> 0.5."""
    assert epsilon < 0.5
    lo, hi = scores.min(), scores.max()
    if hi == lo:
        return np.full_like(scores, 0.5)
    return (scores - lo) / (hi - lo) * (1 - 2 * epsilon) + epsilon


class BaseLabelPropagationRanker:
    def __init__(self, *, knng, nvecs, normalize_scores, sigmoid_before_propagate, calib_a, calib_b,
                 prior_weight, normalize_epsilon=None, **other):
        self.knng = knng
        self.nvecs = nvecs
        self.normalize_scores = normalize_scores
        if normalize_scores:
            assert normalize_epsilon is not None
            self.epsilon = normalize_epsilon
        self.calib_a, self.calib_b = calib_a, calib_b
        self.prior_weight = prior_weight
        self.sigmoid_before_propagate = sigmoid_before_propagate
        self.is_labeled = np.zeros(nvecs)
        self.labels = np.zeros(nvecs)
        self._label_map = {}  # vector id -> label: the labelled set without O(nvecs) scans per round
        self.prior_scores = None
        self._current_scores = None
        self.all_indices = FrozenBitMap(range(nvecs))

    def set_base_scores(self, init_scores):
        assert self.nvecs == init_scores.shape[0]
        if self.normalize_scores:
            init_scores = normalize_scores(init_scores, epsilon=self.epsilon)
        if self.sigmoid_before_propagate:
            self.prior_scores = sigmoid(self.calib_a * (init_scores + self.calib_b))
        else:
            self.prior_scores = init_scores
        # nothing labelled yet: the prior is the score; otherwise propagate right away
        if not self._label_map:
            self._current_scores = self.prior_scores
        else:
            self._current_scores = self._propagate(self.prior_scores)

    def _propagate(self, scores):
        raise NotImplementedError("implement me")

    def update(self, idxs, labels):
        self.update_labels(idxs, labels)
        self.propagate_now()

    def update_labels(self, idxs, labels):
        """the bookkeeping half of update(): record the labels (no propagation yet).  The loops hand over the WHOLE
        labelled set every round (getXy); only the entries that are new or carry another value touch the map."""
        idxs = np.asarray(idxs, dtype=np.int64).reshape(-1)
        labels = np.asarray(labels, dtype=np.float64).reshape(-1)
        # np.isclose(label, 0) or np.isclose(label, 1) (atol 1e-8, rtol 1e-5) for the whole batch at once (exact 0 / 1 first)
        assert ((labels == 0) | (labels == 1)).all() or \
            np.all((np.abs(labels) <= 1e-8) | (np.abs(labels - 1.0) <= 1e-8 + 1e-5))
        was_new = self.is_labeled[idxs] == 0
        changed = was_new | (self.labels[idxs] != labels)
        self.labels[idxs] = labels  # (a repeated id keeps its last label, as the per-item loop did)
        self.is_labeled[idxs] = 1
        if changed.any():
            ci = idxs[changed]
            self._label_map.update(zip(ci.tolist(), self.labels[ci].tolist()))  # the values as assigned: last one wins
            if was_new.any():  # the labelled ids, ascending (== sorted(_label_map) == nonzero(is_labeled))
                self._sorted_ids = np.union1d(self._sorted_label_ids(), idxs[was_new])
            self._labels_stamp = getattr(self, "_labels_stamp", 0) + 1  # invalidates what was derived from the labels

    def _sorted_label_ids(self) -> np.ndarray:
        ids = getattr(self, "_sorted_ids", None)
        if ids is None or ids.shape[0] != len(self._label_map):  # (somebody wrote the map directly)
            ids = self._sorted_ids = np.fromiter(sorted(self._label_map), dtype=np.int64, count=len(self._label_map))
        return ids

    def _refresh_has_negative(self) -> bool:
        stamp = getattr(self, "_labels_stamp", 0)
        if getattr(self, "_neg_stamp", None) != stamp:  # once per label change, not once per call
            self._has_negative = bool((self.labels[self._sorted_label_ids()] == 0).any())
            self._neg_stamp = stamp
        return self._has_negative

    def propagate_now(self):
        """the other half: propagate the recorded labels (PseudoLR runs this beside its pseudo-label draw)"""
        has_negative = self._refresh_has_negative()
        if has_negative:  # the reference skips propagation until a negative label exists
            print(" propagating")
            self._current_scores = self._propagate(self.prior_scores)
        else:
            print(" no negatives yet, skipping propagation")
            self._no_negatives_yet()

    def _no_negatives_yet(self):
        """hook: the scores stay the prior (subclasses may keep them where they are)"""

    def current_scores(self):
        return self._current_scores

    def top_k(self, k, unlabeled_only=True):
        subset = np.where(self.is_labeled < 1)[0] if unlabeled_only else np.arange(self.nvecs)
        raw = self.current_scores()
        order = np.argsort(-raw[subset], kind="stable")[:k]
        top = subset[order]
        return top, raw[top]


class LabelPropagationRanker2(BaseLabelPropagationRanker):
    lp: LabelPropagation

    def __init__(self, *, weight_matrix, verbose: int = 0, device: int = 0, node_order=None, **other):
        super().__init__(knng=None, nvecs=weight_matrix.shape[0], **other)
        self.knng_intra = None
        self.weight_matrix = weight_matrix
        self.lp = LabelPropagation(weight_matrix=weight_matrix, reg_lambda=self.prior_weight, max_iter=300,
                                   verbose=verbose, device=device, node_order=node_order)

    def set_base_scores(self, init_scores):
        super().set_base_scores(init_scores)
        self.lp.set_prior(self.prior_scores)  # constant until the next text query: keep it on the device
        self._resident = False
        if not self._label_map:
            self._no_negatives_yet()

    def _propagate(self, scores):
        ids = self._sorted_label_ids()  # == nonzero(is_labeled)
        vals = self.labels.reshape(-1)[ids]
        if scores is self.prior_scores and self.lp._prior_installed and self.lp.reg_values is self.prior_scores:
            # the loop's case (update(): start == prior): nothing but the labels crosses PCIe; the f64 scores
            # are fetched only if somebody asks for them (current_scores / top_k)
            self.lp.fit_resident(label_ids=ids, label_values=vals)
            self._resident = True
            return None
        self._resident = False
        out = self.lp.fit_transform(label_ids=ids, label_values=vals, reg_values=self.prior_scores, start_value=scores)
        if scores is self.prior_scores:
            self.lp.set_prior(self.prior_scores)  # re-install for the following rounds
        return out

    def can_fuse_round(self) -> bool:
        """True when update() + the next selection can run as ONE device call (LabelPropagation.round): the loop's case
        -- the prior installed on the device is the start iterate of every propagation"""
        return bool(self.lp._prior_installed and self.lp.reg_values is self.prior_scores and self.prior_scores is not None)

    def update_and_select(self, idxs, labels, device_index, *, excluded, k):
        """update(idxs, labels) and, in the same device call, the selection the next next_batch() would ask for:
        the top k distinct non-excluded images by propagated score over the unlabelled vectors (graph_based.py:88-101).
        Follows update()'s rule: no propagation before a negative label exists -- the prior ranks.
        -> (image positions, scores, best rows); the propagated scores stay on the device as after update()."""
        self.update_labels(idxs, labels)
        self._refresh_has_negative()
        ids = self._sorted_label_ids()
        if self._has_negative:
            print(" propagating")
            vals = self.labels.reshape(-1)[ids]
        else:
            print(" no negatives yet, skipping propagation")
            vals = np.zeros(ids.shape[0])
        out = self.lp.round(device_index, propagate=self._has_negative, label_ids=ids, label_values=vals, mask_labeled=True,
                            excluded=excluded, k=k)
        self._resident = True
        self._current_scores = None
        return out

    def _no_negatives_yet(self):
        # the reference serves the prior until a negative label exists; here the prior is in the graph handle already
        # (set_prior): it becomes the resident result with the labelled nodes marked, so these rounds select and
        # re-score on the device like the later ones instead of uploading 12 MB of scores per round
        if self.lp._prior_installed and self.lp.reg_values is self.prior_scores:
            self.lp.prior_as_result(self._sorted_label_ids())
            self._resident = True
            self._current_scores = None

    def current_scores(self):
        if self._current_scores is None and getattr(self, "_resident", False):
            self._current_scores = self.lp.fetch()
        return self._current_scores

    def scores_at(self, rows) -> np.ndarray:
        """current_scores()[rows] without fetching the whole vector when it still lives on the device"""
        if self._current_scores is None and getattr(self, "_resident", False):
            return self.lp.gather(rows)
        return self.current_scores()[rows]

    def scores_on_device(self) -> bool:
        """True when the latest scores live in the label-propagation handle (not yet fetched)"""
        return bool(getattr(self, "_resident", False))
