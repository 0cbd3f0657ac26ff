"""MultiReg (the "seesaw" method) and its RegModule: query-aligned linear scorer with a
database-alignment regulariser w'(X'LX)w, fitted on the labelled tile vectors each round.

Interface of seesaw/loops/multi_reg.py:24-227.  The reference's RegModule is an nn.Module
whose closure torch.optim.LBFGS re-evaluates through autograd; here `fit` hands the
objective to the HIP feedback engine (fused loss/gradient kernels + L-BFGS driver in
libseesaw_hip.so) and only the coefficient vector comes back.
"""
from __future__ import annotations

import math

import numpy as np
import pandas as pd

from .. import _lib
from .._lib import FbObjective
from ..feedback import FeedbackEngine
from .point_based import PointBased

_LOSS_CODE = {"ce_loss": _lib.SSW_FB_LOSS_CE, "pairwise_rank_loss": _lib.SSW_FB_LOSS_PAIRWISE_HINGE,
              "pairwise_logistic_loss": _lib.SSW_FB_LOSS_PAIRWISE_LOGISTIC}


class RegModule:
    def __init__(self, *, dim, xlx_matrix, qvec, label_loss_type, reg_data_lambda, reg_norm_lambda,
                 reg_query_lambda, use_qvec_norm, rank_loss_margin=0.0, pos_weight, verbose=False, max_iter=100,
                 lr=1.0, device: int = 0, engine: FeedbackEngine = None):
        assert label_loss_type in _LOSS_CODE
        qvec = np.asarray(qvec, dtype=np.float32).reshape(-1)
        qn = float(np.linalg.norm(qvec))
        assert not math.isclose(qn, 0.0)
        self.dim = dim
        self.label_loss_type = label_loss_type
        self.qvec = qvec / max(qn, 1e-12)
        self.weight = self.qvec.copy()  # w0 = normalised query (multi_reg.py:39-40)
        self.max_iter, self.lr, self.verbose = int(max_iter), float(lr), verbose
        self.reg_query_lambda, self.reg_norm_lambda, self.reg_data_lambda = reg_query_lambda, reg_norm_lambda, reg_data_lambda
        self.use_qvec_norm = use_qvec_norm
        self.pos_weight = pos_weight
        self.rank_loss_margin = rank_loss_margin
        self._engine = engine or FeedbackEngine(dim, device=device)
        if xlx_matrix is not None and reg_data_lambda != 0:
            self._engine.set_xlx(np.asarray(xlx_matrix, dtype=np.float32))
        self._engine.set_query(self.qvec)
        self.info_ = None

    def _objective(self) -> FbObjective:
        if self.pos_weight == "balanced":
            pw = -1.0
        else:
            assert type(self.pos_weight) is float, "unknown pos weight type"
            pw = self.pos_weight
        return FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=_LOSS_CODE[self.label_loss_type], fit_intercept=0,
                           reg_kind=0, pos_weight=pw, reg_weight=0.0, margin=float(self.rank_loss_margin),
                           reg_norm_lambda=float(self.reg_norm_lambda), reg_data_lambda=float(self.reg_data_lambda),
                           reg_query_lambda=float(self.reg_query_lambda))

    def get_coeff(self):
        w = self.weight
        return (w / max(float(np.linalg.norm(w)), 1e-12)).astype(np.float32)

    def forward(self, X, y=None):
        return np.asarray(X, dtype=np.float32) @ self.weight

    def fit(self, X, y, matchdf, index=None, rows=None):
        """X [n, dim] labelled tile vectors (or index + rows to gather them on the device),
        y [n] in {0,1}, matchdf with a `dbidx` column (sample weight = 1 / #vectors of the image)."""
        n = len(y)
        if n > 0:
            # 1 / (vectors of the same image): matchdf.groupby('dbidx').size() merged back (multi_reg.py:163-165)
            dbidx = matchdf if isinstance(matchdf, np.ndarray) else matchdf.dbidx.values
            if dbidx.shape[0] > 1 and np.all(dbidx[1:] >= dbidx[:-1]):  # an image's tiles are adjacent: run lengths
                cuts = np.flatnonzero(dbidx[1:] != dbidx[:-1]) + 1
                cnt = np.diff(np.concatenate(([0], cuts, [dbidx.shape[0]])))
                vec_weight = np.repeat(1.0 / cnt.astype(np.float64), cnt)
            else:
                _, inv, cnt = np.unique(dbidx, return_inverse=True, return_counts=True)
                vec_weight = 1.0 / cnt[inv].astype(np.float64)
            if X is not None:
                self._engine.set_data(X, center=True)
            else:
                self._engine.set_data_from_index(index, rows, center=True)
            self._engine.set_targets(np.asarray(y, dtype=np.float64), vec_weight)
        else:  # regularisers only (multi_reg.py:171-172 `dl = None`)
            self._engine.set_data(np.zeros((0, self.dim), np.float32), center=False)
            self._engine.set_targets(np.zeros(0))
        try:
            w, info = self._engine.fit(self._objective(), self.weight, max_iter=self.max_iter, lr=self.lr)
        except _lib.SeesawHipError as e:
            raise AssertionError(f"regression training failed: {e}") from e
        assert not np.isnan(w).any()
        self.weight = w
        self.info_ = info
        return [{"k": "total_loss", "loss": info["loss"]}]


class MultiReg(PointBased):
    def __init__(self, gdm, q, params):
        super().__init__(gdm, q, params)
        from .graph_based import get_weight_matrix_from_index
        self.options = self.params.interactive_options
        self.xlx_matrix = None
        if self.options["reg_data_lambda"] > 0:
            self.xlx_matrix = get_weight_matrix_from_index(q.index, self.options["matrix_options"], xlx_matrix=True)
        self._engine = FeedbackEngine(q.index.vectors.shape[1], device=getattr(q.index, "device", 0))

    def set_text_vec(self, tvec):
        super().set_text_vec(tvec)
        # with both regularisers on, optimise the query against them before any label exists
        if self.options["reg_data_lambda"] > 0 and self.options["reg_query_lambda"] > 0 and self.started:
            self.refine()
        else:
            self.curr_vec = self.curr_qvec

    def refine(self, change=None):
        arrays = getattr(self.q, "_matched_arrays", None)
        if arrays is not None and hasattr(self.q.index, "_row_dbidx"):  # getXy()'s columns without the frame
            rows, miou = arrays()
            y = (miou > 0).astype("float")
            matchdf = self.q.index._row_dbidx[rows]
        else:
            matchdf = self.q.getXy()
            rows = matchdf.index.values
            y = matchdf.ys.values
        assert self.curr_qvec is not None
        o = self.options
        model = RegModule(dim=self.q.index.vectors.shape[1], xlx_matrix=self.xlx_matrix, qvec=self.curr_qvec,
                          label_loss_type=o["label_loss_type"], rank_loss_margin=o["rank_loss_margin"],
                          reg_data_lambda=o["reg_data_lambda"], reg_norm_lambda=o["reg_norm_lambda"],
                          use_qvec_norm=o["use_qvec_norm"], reg_query_lambda=o["reg_query_lambda"],
                          verbose=o["verbose"], max_iter=int(o["max_iter"]), pos_weight=o["pos_weight"], lr=o["lr"],
                          engine=self._engine)
        dev = getattr(self.q.index, "_dev", None)
        if dev is not None:  # labelled rows are gathered out of the resident index, no upload
            model.fit(None, y, matchdf, index=dev, rows=rows)
        else:
            model.fit(self.q.index.vectors[rows], y, matchdf)
        self.curr_vec = model.get_coeff()
