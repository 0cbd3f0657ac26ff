"""Dataset / Result / ProbabilityModel of the active-search planners (seesaw/research/active_search/common.py).
Host bookkeeping: which nodes carry a label; the numeric work is in LKNNModel and ssw_lknn_top_sum."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ...bitmap import BitMap, FrozenBitMap


class Dataset:
    """immutable-style view: labels so far (idx2label), the nodes seen, all nodes, and the vectors"""

    def __init__(self, idx2label: dict, seen_indices: BitMap, all_indices: FrozenBitMap, vectors: np.ndarray):
        self.idx2label, self.seen_indices, self.all_indices, self.vectors = idx2label, seen_indices, all_indices, vectors

    @staticmethod
    def from_vectors(vectors):
        return Dataset({}, BitMap(), FrozenBitMap(range(len(vectors))), vectors)

    @staticmethod
    def from_labels(idxs, labels, vectors):
        return Dataset(dict(zip(idxs, labels)), BitMap(idxs), FrozenBitMap(range(len(vectors))), vectors)

    def with_label(self, i, y) -> "Dataset":
        assert i in self.all_indices
        labels = dict(self.idx2label)
        labels[i] = y
        seen = self.seen_indices.copy()
        seen.add(i)
        return Dataset(labels, seen, self.all_indices, self.vectors)

    def get_labels(self):
        idxs = np.array(self.seen_indices)
        return idxs, np.array([self.idx2label[idx] for idx in idxs])

    def remaining_indices(self) -> BitMap:
        return self.all_indices - self.seen_indices


class Result:
    def __init__(self, value: float, index: int, pruned_fraction: Optional[float] = None):
        self.value, self.index, self.pruned_fraction = value, index, pruned_fraction


class ProbabilityModel:
    dataset: Dataset

    def __init__(self, dataset):
        self.dataset = dataset

    def condition(self, idx, y) -> "ProbabilityModel":
        raise NotImplementedError()

    def predict_proba(self, idx: np.ndarray) -> np.ndarray:
        raise NotImplementedError()

    def top_k_remaining(self, top_k: int) -> Tuple[np.ndarray, np.ndarray]:
        idxs = self.dataset.remaining_indices()
        pred = self.predict_proba(idxs)
        order = np.argsort(-pred)[:top_k]
        return np.array([idxs[int(i)] for i in order]), pred[order]

    def probability_bound(self, n) -> float:
        raise NotImplementedError
