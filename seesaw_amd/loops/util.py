"""helpers of the loops (seesaw/loops/util.py:4-33)."""
import os

import numpy as np

from ..nprand import permutation_prefix


def _scores_at(lr, pick):
    """lr.current_scores()[pick] (seesaw/loops/util.py:19); a ranker whose scores are still on the device hands
    back just those entries"""
    at = getattr(lr, "scores_at", None)
    return at(pick) if at is not None else lr.current_scores()[pick]


def makeXy(idx, lr, sample_size, pseudoLabel=True):
    """real labels + a random sample of pseudo-labelled vectors (their propagated scores)."""
    is_labeled = lr.is_labeled > 0
    rows = np.nonzero(is_labeled)[0]
    y = lr.labels[is_labeled]
    is_real = np.ones_like(y)
    if pseudoLabel:
        unl = np.nonzero(~is_labeled)[0]
        pick = unl[permutation_prefix(unl.shape[0], sample_size)]  # == np.random.permutation(n)[:sample_size]
        rows = np.concatenate((rows, pick))
        y = np.concatenate((y, _scores_at(lr, pick)))
        is_real = np.concatenate((is_real, np.zeros(pick.shape[0])))
    return idx.vectors[rows], y, is_real


def makeXy_rows(lr, sample_size):
    """same draw, returning row positions so the vectors can be gathered on the device.  A ranker that keeps its
    labelled set as a map (`_label_map`) spares the three passes over all vectors: the p-th unlabelled row is
    p + #{i : labelled[i] - i <= p} over the sorted labelled rows."""
    label_map = getattr(lr, "_label_map", None)
    if label_map is not None:
        rows = np.fromiter(sorted(label_map), dtype=np.int64, count=len(label_map))  # == nonzero(is_labeled > 0)
        y = lr.labels[rows]
        p = permutation_prefix(lr.is_labeled.shape[0] - rows.shape[0], sample_size)
        pick = p + np.searchsorted(rows - np.arange(rows.shape[0]), p, side="right")  # == nonzero(~is_labeled)[0][p]
    else:
        is_labeled = lr.is_labeled > 0
        rows = np.nonzero(is_labeled)[0]
        y = lr.labels[is_labeled]
        unl = np.nonzero(~is_labeled)[0]
        pick = unl[permutation_prefix(unl.shape[0], sample_size)]  # == np.random.permutation(n)[:sample_size]
    is_real = np.ones_like(y)
    return (np.concatenate((rows, pick)), np.concatenate((y, _scores_at(lr, pick))),
            np.concatenate((is_real, np.zeros(pick.shape[0]))))


def get_image_paths(image_root, path_array, idxs):
    return [os.path.normpath(f"{image_root}/{path_array[int(i)]}").replace("//", "/") for i in idxs]


def clean_path(path):
    return os.path.normpath(os.path.abspath(os.path.realpath(path)))
