"""Rank-and-loss update of the early seesaw loop (seesaw/pairwise_rank_loss.py): `compute_inversions` (:24-43),
`RankAndLoss` forward / backward (:46-135), `RankLoss` (:138-156) and the online `VecState` (:159-187).

RankAndLoss is the pairwise hinge between positives and negatives with the margin taken off the positive scores,
normalised by n_pos x n_neg; its backward is X' (inversions x sign / npairs).  Both come out of the pairwise
kernel the MultiReg fit uses (ssw_rank_pairwise): with binary targets every (positive, negative) pair appears in
two columns, so sum of columns = 2 x the loss numerator and d(sum)/d scores = 2 x (inversions x sign)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .rank_loss import max_inversions, pairwise_sums


def compute_inversions(labs: np.ndarray, scores: np.ndarray, device: int = 0) -> np.ndarray:
    """per item, in descending score order: a positive counts the negatives before it, a negative the positives
    after it (ties broken by index; the reference's np.argsort leaves them unspecified)"""
    assert labs.shape == scores.shape and labs.ndim == 1
    lab = np.ascontiguousarray(labs.astype(bool), dtype=np.uint8)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    out = np.zeros(lab.shape[0], dtype=np.int64)
    _lib.call("ssw_rank_inversions", int(device), ctypes.c_void_p(lab.ctypes.data), ctypes.c_void_p(s.ctypes.data),
              lab.shape[0], ctypes.c_void_p(out.ctypes.data))
    return out


def rank_and_loss(weight: np.ndarray, inputs: np.ndarray, labels: np.ndarray, margin: float, device: int = 0):
    """RankAndLoss.apply(weight, inputs, labels, margin) and its backward in one call -> (loss, d loss / d weight)"""
    w = np.asarray(weight, dtype=np.float32).reshape(-1)
    X = np.asarray(inputs, dtype=np.float32)
    lab = (np.asarray(labels).reshape(-1) == 1.0)
    npos, nneg = int(lab.sum()), int((~lab).sum())
    npairs = npos * nneg
    if npairs == 0:
        return 0.0, np.zeros_like(w)
    scores = (X @ w).astype(np.float32)
    scores[lab] -= np.float32(margin)  # the margin comes off the positive scores
    t = lab.astype(np.float32)
    item, grad_scores = pairwise_sums(t, scores=scores, margin=0.0, logistic=False, coef=max_inversions(t), device=device)
    loss = float(item.sum()) / (2.0 * npairs)
    if loss == 0.0:
        return 0.0, np.zeros_like(w)
    coeffs = grad_scores.astype(np.float32) / np.float32(2.0 * npairs)
    return loss, (X.T @ coeffs).astype(np.float32)


class RankLoss:
    """holds the (initially normalised) vector; forward returns the loss and keeps the gradient"""

    def __init__(self, w: np.ndarray, margin: float, dummy: bool = False, device: int = 0):
        w = np.asarray(w, dtype=np.float32).reshape(-1)
        self.w = w / max(float(np.linalg.norm(w)), 1e-12)
        self.margin, self.dummy, self.device = float(margin), dummy, device
        self.grad = np.zeros_like(self.w)

    def forward(self, dat, labels):
        loss, self.grad = rank_and_loss(self.w, dat, labels, self.margin, device=self.device)
        return (1.0 if loss != 0.0 else 0.0) if self.dummy else loss

    __call__ = forward


class VecState:
    """one optimiser step per `update` (the reference is only ever constructed with torch.optim.SGD: old_seesaw.py:25-30)"""

    def __init__(self, w: np.ndarray, margin: float, opt_class=None, opt_params=None, renormalize=False, device: int = 0):
        opt_params = dict(opt_params or {})
        name = getattr(opt_class, "__name__", "SGD") if opt_class is not None else "SGD"
        if name != "SGD" or any(opt_params.get(k) for k in ("momentum", "weight_decay", "nesterov", "dampening")):
            raise NotImplementedError("VecState steps with plain SGD, as the reference's only caller does")
        self.lr = float(opt_params.get("lr", 1e-3))
        self.mod = RankLoss(w, margin, dummy=True, device=device)
        self.renormalize = renormalize

    def get_vec(self):
        return self.mod.w.copy()

    def update(self, vecs, labels):
        self.mod(np.asarray(vecs, dtype=np.float32), (np.asarray(labels) == 1).astype(np.float32))
        self.mod.w = (self.mod.w - np.float32(self.lr) * self.mod.grad).astype(np.float32)
        if self.renormalize:
            self.mod.w = self.mod.w / max(float(np.linalg.norm(self.mod.w)), 1e-12)
