"""FeedbackEngine: ctypes front of the ssw_fb_* entry points (fused loss+gradient kernels and
the L-BFGS driver in libseesaw_hip.so).  Used by seesaw_amd.logistic_regression and
seesaw_amd.loops.multi_reg, which keep the reference's class interfaces."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib
from ._lib import FbObjective


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


_POOL = {}  # (device, dim) -> idle engines
_POOL_MAX = 4


class FeedbackEngine:
    @classmethod
    def acquire(cls, dim: int = 512, device: int = 0) -> "FeedbackEngine":
        """an idle engine of that shape, wiped (ssw_fb_reset), or a new one.  The loops build a fresh scorer object every
        refine as the reference does; giving each its own device allocations cost 1.5-3 ms per round in hipMalloc /
        hipFree alone."""
        idle = _POOL.get((int(device), int(dim)))
        if idle:
            eng = idle.pop()
            _lib.call("ssw_fb_reset", eng._h)
            eng.n = 0
            return eng
        return cls(dim, device)

    def release(self):
        """hand the engine back for the next acquire (closed when the pool is full)"""
        if not self._h:
            return
        idle = _POOL.setdefault((self.device, self.dim), [])
        if len(idle) < _POOL_MAX:
            idle.append(self)
        else:
            self.close()

    def __init__(self, dim: int = 512, device: int = 0):
        self.dim = int(dim)
        self.device = int(device)
        self._h = ctypes.c_void_p()
        self.n = 0
        _lib.call("ssw_fb_create", self.device, self.dim, ctypes.byref(self._h))

    def close(self):
        if self._h:
            _lib.load().ssw_fb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data -------------------------------------------------------------------------
    def set_data(self, X: np.ndarray, center: bool):
        X = np.ascontiguousarray(X, dtype=np.float32)
        assert X.ndim == 2 and X.shape[1] == self.dim
        self.n = X.shape[0]
        _lib.call("ssw_fb_set_data", self._h, _p(X), self.n, int(center))

    def set_data_from_index(self, device_index, rows: np.ndarray, center: bool):
        """gather the labelled rows out of the index matrix already resident in HBM."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        vec_ptr, _ = device_index.device_ptrs()
        self.n = rows.shape[0]
        _lib.call("ssw_fb_set_data_from_device", self._h, ctypes.c_void_p(vec_ptr), device_index.n_rows,
                  _p(rows), self.n, int(center))

    def set_pseudo_sample(self, device_index, dev_scores_ptr: int, labelled_rows: np.ndarray, labelled_y: np.ndarray,
                          drawn: np.ndarray, real_weight: float, center: bool):
        """PseudoLR's training set put together on the device (ssw_fb_set_pseudo_sample): the labelled rows, then the
        drawn-th unlabelled rows with their propagated scores as targets; returns the number of rows"""
        rows = np.ascontiguousarray(labelled_rows, dtype=np.int64)
        y = np.ascontiguousarray(labelled_y, dtype=np.float32)
        drawn = np.ascontiguousarray(drawn, dtype=np.int64)
        assert rows.shape == y.shape
        vec_ptr, _ = device_index.device_ptrs()
        self.n = rows.shape[0] + drawn.shape[0]
        _lib.call("ssw_fb_set_pseudo_sample", self._h, ctypes.c_void_p(vec_ptr), device_index.n_rows,
                  ctypes.c_void_p(dev_scores_ptr), _p(rows), _p(y), rows.shape[0], _p(drawn), drawn.shape[0],
                  float(real_weight), int(center))
        return self.n

    def set_targets(self, y: np.ndarray, sample_weight: Optional[np.ndarray] = None):
        y = np.ascontiguousarray(np.asarray(y).reshape(-1), dtype=np.float32)
        assert y.shape[0] == self.n
        sw = None
        if sample_weight is not None:
            sw = np.ascontiguousarray(np.asarray(sample_weight).reshape(-1), dtype=np.float32)
            assert sw.shape[0] == self.n
        _lib.call("ssw_fb_set_targets", self._h, _p(y), _p(sw))

    def set_query(self, q: np.ndarray):
        q = np.ascontiguousarray(np.asarray(q).reshape(-1), dtype=np.float32)
        assert q.shape[0] == self.dim
        _lib.call("ssw_fb_set_query", self._h, _p(q))

    def set_xlx(self, xlx: np.ndarray):
        xlx = np.ascontiguousarray(xlx, dtype=np.float32)
        assert xlx.shape == (self.dim, self.dim)
        _lib.call("ssw_fb_set_xlx", self._h, _p(xlx))

    def mean(self) -> np.ndarray:
        mu = np.empty(self.dim, dtype=np.float32)
        _lib.call("ssw_fb_get_mean", self._h, _p(mu))
        return mu

    # ---- objective --------------------------------------------------------------------
    def lossgrad(self, obj: FbObjective, w: np.ndarray):
        P = self.dim + (1 if (obj.kind == _lib.SSW_FB_LOGREG and obj.fit_intercept) else 0)
        w = np.ascontiguousarray(np.asarray(w).reshape(-1), dtype=np.float32)
        assert w.shape[0] == P
        loss = ctypes.c_float(0)
        grad = np.empty(P, dtype=np.float32)
        parts = np.empty(4, dtype=np.float32)
        _lib.call("ssw_fb_lossgrad", self._h, ctypes.byref(obj), _p(w), ctypes.byref(loss), _p(grad), _p(parts))
        return float(loss.value), grad, parts

    def scores(self, w: np.ndarray, has_bias: bool = False) -> np.ndarray:
        w = np.ascontiguousarray(np.asarray(w).reshape(-1), dtype=np.float32)
        out = np.empty(self.n, dtype=np.float32)
        _lib.call("ssw_fb_scores", self._h, _p(w), int(has_bias), _p(out))
        return out

    def fit(self, obj: FbObjective, w0: np.ndarray, max_iter: int, lr: float = 1.0):
        """one LBFGS(max_iter, lr, strong_wolfe).step(closure) from w0; returns (w, info)."""
        w = np.array(np.asarray(w0).reshape(-1), dtype=np.float32)
        iters, evals, loss = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_float(0)
        _lib.call("ssw_fb_fit", self._h, ctypes.byref(obj), _p(w), int(max_iter), float(lr),
                  ctypes.byref(iters), ctypes.byref(evals), ctypes.byref(loss))
        on_dev = ctypes.c_int32(0)
        _lib.call("ssw_fb_last_fit_on_device", self._h, ctypes.byref(on_dev))
        return w, {"n_iter": iters.value, "func_evals": evals.value, "loss": float(loss.value),
                   "on_device": bool(on_dev.value)}

    # ---- two linear outputs (MultiRegModule, loops/multi_reg_neg.py) ---------------------
    def set_targets2(self, y2: np.ndarray, sample_weight: Optional[np.ndarray] = None):
        y2 = np.ascontiguousarray(np.asarray(y2), dtype=np.float32)
        assert y2.shape == (self.n, 2)
        sw = None
        if sample_weight is not None:
            sw = np.ascontiguousarray(np.asarray(sample_weight).reshape(-1), dtype=np.float32)
            assert sw.shape[0] == self.n
        _lib.call("ssw_fb_set_targets2", self._h, _p(y2), _p(sw))

    def lossgrad2(self, W: np.ndarray, reg_norm_lambda: float, reg_query_lambda: float):
        W = np.ascontiguousarray(np.asarray(W), dtype=np.float32)
        assert W.shape == (2, self.dim)
        loss = ctypes.c_float(0)
        grad = np.empty((2, self.dim), dtype=np.float32)
        parts = np.empty(5, dtype=np.float32)
        _lib.call("ssw_fb_lossgrad2", self._h, _p(W), float(reg_norm_lambda), float(reg_query_lambda),
                  ctypes.byref(loss), _p(grad), _p(parts))
        return float(loss.value), grad, parts

    def fit2(self, W0: np.ndarray, reg_norm_lambda: float, reg_query_lambda: float, max_iter: int, lr: float = 1.0):
        W = np.array(np.asarray(W0), dtype=np.float32)
        assert W.shape == (2, self.dim)
        iters, evals, loss = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_float(0)
        _lib.call("ssw_fb_fit2", self._h, _p(W), float(reg_norm_lambda), float(reg_query_lambda), int(max_iter), float(lr),
                  ctypes.byref(iters), ctypes.byref(evals), ctypes.byref(loss))
        return W, {"n_iter": iters.value, "func_evals": evals.value, "loss": float(loss.value), "on_device": False}
