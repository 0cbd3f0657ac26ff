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


def draw_unlabelled(lr, sample_size, device=None):
    """np.random.permutation(#unlabelled)[:sample_size] for a ranker that keeps its labelled set as a map: depends on the
    labels recorded so far, not on the propagation -- PseudoLR makes it while the propagation runs"""
    return permutation_prefix(lr.is_labeled.shape[0] - len(lr._label_map), sample_size, device=device)


def makeXy_rows(lr, sample_size, drawn=None):
    """same draw, returning row positions so the vectors can be gathered on the device.  A ranker that keeps its
    labelled set as a map (`_label_map`) spares the three passes over all vectors: the p-th unlabelled row is
    p + #{i : labelled[i] - i <= p} over the sorted labelled rows.  `drawn`: the permutation prefix, when the caller
    has made the draw already (draw_unlabelled)."""
    label_map = getattr(lr, "_label_map", None)
    if label_map is not None:
        rows = lr._sorted_label_ids() if hasattr(lr, "_sorted_label_ids") else \
            np.fromiter(sorted(label_map), dtype=np.int64, count=len(label_map))  # == nonzero(is_labeled > 0)
        y = lr.labels[rows]
        if drawn is None:
            # a ranker whose scores live on a GPU lends it to the draw's walk through the swaps (nprand.py)
            dev = getattr(getattr(lr, "lp", None), "device", None) if getattr(lr, "scores_on_device", lambda: False)() else None
            drawn = permutation_prefix(lr.is_labeled.shape[0] - rows.shape[0], sample_size, device=dev)
        p = drawn
        # == nonzero(~is_labeled)[0][p]: the p-th unlabelled row sits #{j : rows[j] - j <= p} places further on
        n_unl = lr.is_labeled.shape[0] - rows.shape[0]
        if n_unl <= 64 * max(int(p.shape[0]), 1):  # small index: a step table over the unlabelled positions and one gather
            step = np.bincount(rows - np.arange(rows.shape[0]), minlength=n_unl + 1)[:n_unl + 1]
            pick = p + np.cumsum(step)[p]
        else:
            pick = p + np.searchsorted(rows - np.arange(rows.shape[0]), p, side="right")
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
