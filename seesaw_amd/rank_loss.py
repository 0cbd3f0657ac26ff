"""Pairwise rank losses on the GPU behind the reference's function names (seesaw/rank_loss.py).

`ref_pairwise_rank_loss` / `ref_pairwise_logistic_loss` with aggregate='sum' and
`ref_pairwise_rank_loss_gradient` go through ssw_rank_pairwise -- the very kernel the MultiReg fit evaluates
every closure call (csrc/feedback.hip, k_fb_pairwise)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))


def max_inversions(target) -> np.ndarray:
    """per item: how many other items carry a different target (rank_loss.py:56,90)"""
    t = _f32(target)
    _, inv, counts = np.unique(t, return_inverse=True, return_counts=True)
    return (t.shape[0] - counts[inv]).astype(np.float32)


def pairwise_sums(target, *, scores, margin=0.0, logistic=False, coef=None, device: int = 0):
    """-> (item_loss f64 [n], grad f32 [n]): item_loss[j] = coef_j / max_inversions_j * sum_i loss_ij and
    grad = d(sum_j item_loss_j) / d scores.  coef None = ones (RegModule's normalised form)."""
    t, s = _f32(target), _f32(scores)
    assert t.shape == s.shape
    n = t.shape[0]
    c = None if coef is None else _f32(coef)
    item = np.zeros(n, dtype=np.float64)
    grad = np.zeros(n, dtype=np.float32)
    _lib.call("ssw_rank_pairwise", int(device), int(bool(logistic)), ctypes.c_void_p(t.ctypes.data),
              ctypes.c_void_p(s.ctypes.data), None if c is None else ctypes.c_void_p(c.ctypes.data), n, float(margin),
              ctypes.c_void_p(item.ctypes.data), ctypes.c_void_p(grad.ctypes.data))
    return item, grad


def ref_pairwise_rank_loss(target, *, scores, margin, aggregate="sum", return_max_inversions=False, device: int = 0):
    """hinge: sum_i max(0, margin - t_ij s_ij) - margin [t_ij == 0] per column j (rank_loss.py:63-95)"""
    assert aggregate == "sum", "the n x n matrix form is not materialised on the GPU"
    mx = max_inversions(target)
    item, _ = pairwise_sums(target, scores=scores, margin=margin, logistic=False, coef=mx, device=device)
    return (item, mx) if return_max_inversions else item


def ref_pairwise_logistic_loss(target, *, scores, aggregate="sum", return_max_inversions=False, device: int = 0):
    """logistic: sum_i t_ij^2 log(1 + exp(-t_ij s_ij)) per column j (rank_loss.py:34-61)"""
    assert aggregate == "sum", "the n x n matrix form is not materialised on the GPU"
    mx = max_inversions(target)
    item, _ = pairwise_sums(target, scores=scores, logistic=True, coef=mx, device=device)
    return (item, mx) if return_max_inversions else item


def ref_pairwise_rank_loss_gradient(target, *, scores, margin, device: int = 0):
    """d(sum of the hinge column sums) / d scores (rank_loss.py:98-106)"""
    _, grad = pairwise_sums(target, scores=scores, margin=margin, logistic=False, coef=max_inversions(target),
                            device=device)
    return grad


def quick_pairwise_gradient_zero_margin(target, *, scores, return_max_inversions=False, device: int = 0):
    """gradient of the zero-margin pairwise rank loss in O(n log n) in the reference (two lexicographic sorts,
    rank_loss.py:109-161); here ssw_rank_quick_gradient counts the two ranks directly -- same integers, ties
    included.  -> 2 x net position change [, max_reversals, total_pairs]"""
    t, s = _f32(target), _f32(scores)
    assert t.shape == s.shape
    n = t.shape[0]
    grad = np.zeros(n, dtype=np.float32)
    maxrev = np.zeros(n, dtype=np.float32)
    total = ctypes.c_int64(0)
    _lib.call("ssw_rank_quick_gradient", int(device), ctypes.c_void_p(t.ctypes.data), ctypes.c_void_p(s.ctypes.data), n,
              ctypes.c_void_p(grad.ctypes.data), ctypes.c_void_p(maxrev.ctypes.data), ctypes.byref(total))
    return (grad, maxrev, int(total.value)) if return_max_inversions else grad


def cheap_pairwise_rank_loss(target, *, scores, normalized=True, device: int = 0):
    """_CheapPairwiseRankingLoss (rank_loss.py:164-187): forward |gradient| x factor and the backward it hands to
    autograd, gradient x factor (x the incoming gradient).  -> (per-item loss [n], d(sum of the losses) / d scores [n])"""
    grad, _, total = quick_pairwise_gradient_zero_margin(target, scores=scores, return_max_inversions=True, device=device)
    factor = np.float32(1.0) if not normalized else (np.float32(1.0) / np.float32(total) if total else np.float32(np.inf))
    return np.abs(grad) * factor, grad * factor
