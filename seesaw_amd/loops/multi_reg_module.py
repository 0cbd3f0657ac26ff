"""MultiRegModule: the two-output scorer of the multi_reg_neg loop (seesaw/loops/multi_reg_module.py:40-165).

Row 0 of the weight matrix is the target query, row 1 the "confusion" class the user marked as not-the-target; the
logits use the L2-normalised rows.  The reference is an nn.Module whose closure torch.optim.LBFGS re-evaluates through
autograd; here `fit` hands the objective to the HIP feedback engine (ssw_fb_fit2) and the weights come back.
"""
from __future__ import annotations

import math

import numpy as np

from .. import _lib
from ..feedback import FeedbackEngine


def _linear_init(dim: int) -> np.ndarray:
    """nn.Linear(dim, 2, bias=False)'s start weights (multi_reg_module.py:50): drawn from torch's global generator the
    way the reference draws them, so a caller that seeds torch gets the reference's start point"""
    import torch
    return torch.nn.Linear(in_features=dim, out_features=2, bias=False).weight.detach().numpy().copy()


def _dataloader_draws(n: int):
    """The reference iterates a DataLoader(batch_size=n, shuffle=True) once per fit (multi_reg_module.py:155): that
    takes numbers from torch's global generator (the sampler's seed, the loader's base seed), so the NEXT fit's
    nn.Linear start weights depend on it.  Iterating the same loader here keeps a session on the reference's generator
    stream under the same torch seed; the row order itself is irrelevant to the sums (they are order-free up to f32
    rounding) and is not used."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    for _ in DataLoader(TensorDataset(torch.zeros(n, 1)), batch_size=n, shuffle=True):
        pass


class MultiRegModule:
    def __init__(self, *, qvec, qvec2=None, reg_norm_lambda, reg_query_lambda, verbose=False, max_iter=100, lr=1.0,
                 dim: int = None, device: int = 0, engine: FeedbackEngine = None, weight0: np.ndarray = None):
        qvec = np.asarray(qvec, dtype=np.float32).reshape(-1)
        qn = float(np.linalg.norm(qvec))
        assert not math.isclose(qn, 0.0)
        self.dim = int(dim or qvec.shape[0])
        self.qvec = qvec / max(qn, 1e-12)
        self.weight = np.asarray(weight0, dtype=np.float32).copy() if weight0 is not None else _linear_init(self.dim)
        assert self.weight.shape == (2, self.dim)
        self.max_iter, self.lr, self.verbose = int(max_iter), float(lr), verbose
        self.reg_query_lambda, self.reg_norm_lambda = float(reg_query_lambda), float(reg_norm_lambda)
        self._engine = engine or FeedbackEngine(self.dim, device=device)
        self._engine.set_query(self.qvec)
        self.info_ = None

    def get_coeff(self):
        w = self.weight[0]
        return (w / max(float(np.linalg.norm(w)), 1e-12)).astype(np.float32)

    def get_confusion_coeff(self):
        w = self.weight[1]
        return (w / max(float(np.linalg.norm(w)), 1e-12)).astype(np.float32)

    def forward(self, X, y=None):
        return np.asarray(X, dtype=np.float32) @ self.weight.T

    def _install(self, X, y, matchdf, index=None, rows=None):
        y = np.asarray(y, dtype=np.float32)
        assert y.ndim == 2 and y.shape[1] == 2  # target, confusion class (multi_reg_module.py:83)
        # 1 / (vectors of the same image): matchdf.groupby('dbidx').size() merged back (multi_reg_module.py:147-149)
        _, inv, cnt = np.unique(matchdf.dbidx.values, return_inverse=True, return_counts=True)
        vec_weight = 1.0 / cnt[inv].astype(np.float64)
        if X is not None:
            self._engine.set_data(X, center=True)
        else:
            self._engine.set_data_from_index(index, rows, center=True)
        self._engine.set_targets2(y, vec_weight)

    def lossgrad(self, W=None):
        """one closure evaluation on the installed rows (tests / diagnostics)"""
        return self._engine.lossgrad2(self.weight if W is None else W, self.reg_norm_lambda, self.reg_query_lambda)

    def fit(self, X, y, matchdf, index=None, rows=None):
        n = len(y)
        if n == 0:  # the reference's _step has no label term to evaluate without rows and raises (multi_reg_module.py:110)
            raise AssertionError("MultiRegModule.fit needs at least one labelled vector")
        self._install(X, y, matchdf, index=index, rows=rows)
        _dataloader_draws(n)
        try:
            W, info = self._engine.fit2(self.weight, self.reg_norm_lambda, self.reg_query_lambda, max_iter=self.max_iter,
                                        lr=self.lr)
        except _lib.SeesawHipError as e:
            raise AssertionError(f"regression training failed: {e}") from e
        assert not np.isnan(W).any()
        self.weight = W
        self.info_ = info
        return [{"k": "loss", "loss": info["loss"]}]
