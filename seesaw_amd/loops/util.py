"""helpers of the loops (seesaw/loops/util.py:4-33)."""
import os

import numpy as np

from ..nprand import permutation_prefix


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
        y = np.concatenate((y, lr.current_scores()[pick]))
        is_real = np.concatenate((is_real, np.zeros(pick.shape[0])))
    return idx.vectors[rows], y, is_real


def makeXy_rows(lr, sample_size):
    """same draw, returning row positions so the vectors can be gathered on the device."""
    is_labeled = lr.is_labeled > 0
    rows = np.nonzero(is_labeled)[0]
    y = lr.labels[is_labeled]
    is_real = np.ones_like(y)
    unl = np.nonzero(~is_labeled)[0]
    pick = unl[permutation_prefix(unl.shape[0], sample_size)]  # == np.random.permutation(n)[:sample_size]
    return (np.concatenate((rows, pick)), np.concatenate((y, lr.current_scores()[pick])),
            np.concatenate((is_real, np.zeros(pick.shape[0]))))


def get_image_paths(image_root, path_array, idxs):
    return [os.path.normpath(f"{image_root}/{path_array[int(i)]}").replace("//", "/") for i in idxs]


def clean_path(path):
    return os.path.normpath(os.path.abspath(os.path.realpath(path)))
