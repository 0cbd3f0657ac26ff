"""LogisticRegressionPT: linear relevance scorer with a query-vector regulariser, fitted by
L-BFGS -- the reference's interface (seesaw/logistic_regression.py:270-421) over the HIP
feedback engine (loss + gradient kernels and the L-BFGS driver in libseesaw_hip.so)."""
from __future__ import annotations

import math

import numpy as np

from . import _lib
from ._lib import FbObjective
from .feedback import FeedbackEngine


def _default_linear_init(dim: int, fit_intercept: bool):
    """nn.Linear(dim, 1)'s default initialisation (what LogisticRegModule starts from,
    logistic_regression.py:71), drawn from torch's global generator so torch.manual_seed
    governs it exactly as in the reference."""
    import torch
    # nn.Linear.reset_parameters (torch/nn/modules/linear.py): the same two calls on bare tensors -- the same draws from the
    # generator without building a Module per fit
    w = torch.empty(1, dim)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    out = w.numpy().reshape(-1)
    if fit_intercept:
        b = torch.empty(1)
        bound = 1 / math.sqrt(dim) if dim > 0 else 0
        torch.nn.init.uniform_(b, -bound, bound)
        out = np.concatenate([out, b.numpy()])
    return out.astype(np.float32, copy=True)


class LogisticRegressionPT:
    def __init__(self, *, class_weights, scale, reg_lambda, regularizer_vector, fit_intercept, verbose=False,
                 max_iter=100, lr=1.0, device: int = 0, **kwargs):
        """regularizer_vector: ndarray -> pull the direction towards that vector; 'norm' /
        'norm1' -> only a norm penalty; None -> no regulariser."""
        assert scale in ["centered", None]
        self.class_weights = class_weights
        self.scale = scale
        self.reg_lambda = reg_lambda
        self.verbose = verbose
        self.fit_intercept = bool(fit_intercept)
        self.max_iter = int(max_iter)
        self.lr = float(lr)
        self.kwargs = kwargs
        self._mu = None          # column means of the last fit's rows, fetched from the device when somebody asks (mu_)
        self.coef_ = None        # [dim] (+1 with intercept) f32 -- the fitted parameters
        self.losses_ = None
        self.regularizer_vector = None
        if isinstance(regularizer_vector, np.ndarray):
            v = regularizer_vector.reshape(-1).astype(np.float32)
            self.regularizer_vector = v / max(float(np.linalg.norm(v)), 1e-12)
            self.regularization_type = "vector"
        elif isinstance(regularizer_vector, str):
            assert regularizer_vector in ("norm", "norm1")
            self.regularization_type = regularizer_vector
        else:
            assert regularizer_vector is None
            self.regularization_type = None
        self._device = device
        self._engine = None
        self._dim = None

    @property
    def mu_(self):
        """the column means the rows were centred by (StandardScaler(with_std=False).mean_, logistic_regression.py:299-300)"""
        if self._mu is None and self._engine is not None:
            self._mu = self._engine.mean()
        return self._mu

    @mu_.setter
    def mu_(self, value):
        self._mu = value

    # ---- helpers ------------------------------------------------------------------------
    def _objective(self, n_examples: int, pos_weight: float) -> FbObjective:
        reg_kind = {None: _lib.SSW_FB_REG_NONE, "vector": _lib.SSW_FB_REG_VECTOR, "norm": _lib.SSW_FB_REG_NORM,
                    "norm1": _lib.SSW_FB_REG_NORM1}[self.regularization_type]
        return FbObjective(kind=_lib.SSW_FB_LOGREG, loss_type=0, fit_intercept=int(self.fit_intercept),
                           reg_kind=reg_kind, pos_weight=float(pos_weight),
                           reg_weight=float(self.reg_lambda) / n_examples, margin=0.0,
                           reg_norm_lambda=0.0, reg_data_lambda=0.0, reg_query_lambda=0.0)

    def __del__(self):
        eng, self._engine = getattr(self, "_engine", None), None
        if eng is not None:
            try:
                eng.release()
            except Exception:
                pass

    def _ensure_engine(self, dim: int):
        if self._engine is None or self._dim != dim:
            if self._engine is not None:
                self._engine.release()
            self._engine = FeedbackEngine.acquire(dim, device=self._device)
            self._dim = dim
            if self.regularizer_vector is not None:
                self._engine.set_query(self.regularizer_vector)

    # ---- reference interface ------------------------------------------------------------
    def fit(self, X, y, sample_weights=None, w0: np.ndarray = None, index=None, rows=None, pseudo=None):
        """X [n, dim] (or `index` + `rows` to gather the vectors on the device), y [n] or [n,1].
        pseudo = (dev_scores_ptr, labelled_rows, labelled_y, drawn, real_weight) with `index`: PseudoLR's set assembled on
        the device (FeedbackEngine.set_pseudo_sample) -- X, y, sample_weights and rows are not used then."""
        dim = X.shape[1] if X is not None else index.dim
        self._ensure_engine(dim)
        center = self.scale == "centered"
        if pseudo is not None:
            if self.class_weights == "balanced":  # (not an assert: python -O must not let it through)
                raise ValueError("class_weights='balanced' needs the targets on the host; a device-assembled pseudo-sample has none")
            n_examples = self._engine.set_pseudo_sample(index, *pseudo, center=center)
        else:
            y = np.asarray(y, dtype=np.float64).reshape(-1)
            n_examples = y.shape[0]
            if X is not None:
                self._engine.set_data(X, center=center)
            else:
                self._engine.set_data_from_index(index, rows, center=center)
        self._mu = None  # (a 2-KB copy back and a stream wait per fit; the loops only read get_coeff())
        if self.class_weights == "balanced":
            npos, nneg = int((y == 1).sum()), int((y == 0).sum())
            pos_weight = max(nneg, 1) / max(npos, 1)
        else:
            pos_weight = float(self.class_weights)
        if self.coef_ is None:
            start = _default_linear_init(dim, self.fit_intercept) if w0 is None else np.asarray(w0, np.float32).reshape(-1)
        else:  # warm start, as the reference (which has not implemented it for 'balanced')
            assert self.class_weights != "balanced", "implement this case"
            start = self.coef_
        if pseudo is None:
            self._engine.set_targets(y, sample_weights)
        obj = self._objective(n_examples, pos_weight)
        try:
            w, info = self._engine.fit(obj, start, max_iter=self.max_iter, lr=self.lr)
        except _lib.SeesawHipError as e:
            if e.status == -5:
                raise ValueError("regression training failed with a nan") from e
            raise
        if math.isnan(info["loss"]) or math.isinf(info["loss"]):
            raise ValueError("regression training failed with a nan")
        self.coef_ = w
        self.losses_ = [{"k": "total_loss", "loss": info["loss"]}]
        self.info_ = info
        if self.verbose:
            print(f"regression converged after {info['n_iter']} iterations. final_loss={info['loss']}")

    def get_coeff(self):
        assert self.coef_ is not None
        return self.coef_[: self._dim].reshape(1, -1).copy()

    def get_intercept(self):
        b = self.coef_[self._dim] if self.fit_intercept else 0.0
        return np.array([-(self.coef_[: self._dim] @ self.mu_) + b], dtype=np.float32)

    def predict_proba(self, X):
        X = np.asarray(X, dtype=np.float32)
        if self.scale == "centered":
            X = X - self.mu_.reshape(1, -1)
        z = X @ self.coef_[: self._dim] + (self.coef_[self._dim] if self.fit_intercept else 0.0)
        return (1.0 / (1.0 + np.exp(-z))).reshape(-1, 1).astype(np.float32)


class RankRegressionPT(LogisticRegressionPT):
    """linear scorer fitted on the cheap pairwise rank loss (seesaw/logistic_regression.py:126-267 over
    RankingRegModule :16-65): sum_i |g_i| / total_pairs + (reg_lambda / n) R(w), g the zero-margin pairwise gradient;
    the regulariser options and get_coeff / predict_proba are LogisticRegressionPT's.  One L-BFGS step per fit, the
    "gradient" being what _CheapPairwiseRankingLoss.backward returns (g / total_pairs through X)."""

    def __init__(self, scale, reg_lambda, regularizer_vector, verbose=False, max_iter=100, lr=1.0, device: int = 0, **kwargs):
        super().__init__(class_weights=1.0, scale=scale, reg_lambda=reg_lambda, regularizer_vector=regularizer_vector,
                         fit_intercept=False, verbose=verbose, max_iter=max_iter, lr=lr, device=device, **kwargs)

    def _objective(self, n_examples: int, pos_weight: float) -> FbObjective:
        obj = super()._objective(n_examples, pos_weight)
        obj.kind = _lib.SSW_FB_RANKREG
        return obj

    def fit(self, X, y, sample_weights=None, w0: np.ndarray = None, index=None, rows=None):
        if sample_weights is not None:
            raise NotImplementedError("handle weights later on")  # as the reference (logistic_regression.py:37)
        return super().fit(X, y, None, w0=w0, index=index, rows=rows)

    def lossgrad(self, w):
        """one closure evaluation at w on the installed data (for tests / diagnostics)"""
        return self._engine.lossgrad(self._objective(self._engine.n, 1.0), w)

    def predict_proba(self, X):
        """RankingRegModule.forward is the raw linear score (no sigmoid)"""
        X = np.asarray(X, dtype=np.float32)
        if self.scale == "centered":
            X = X - self.mu_.reshape(1, -1)
        return (X @ self.coef_[: self._dim]).reshape(-1, 1).astype(np.float32)
