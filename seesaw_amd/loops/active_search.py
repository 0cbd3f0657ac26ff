"""ActiveSearch / LKNNSearch loops (seesaw/loops/active_search.py:32-223): the next image is chosen by planning over
the L-KNN model instead of by the current query vector.  ActiveSearch plans with efficient non-myopic search (the
vectorised two-step look-ahead runs on the GPU); LKNNSearch greedily takes the most probable remaining node."""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp

from ..calibration import FixedCalibrator
from ..research.active_search.common import Dataset
from ..research.active_search.efficient_nonmyopic_search import efficient_nonmyopic_search
from .graph_based import get_label_prop
from .LKNN_model import LKNNModel, initial_gamma_array
from .loop_base import LoopBase


def _vector_position(index, dbidx) -> int:
    """first vector of image `dbidx` (`vector_meta.query(f'dbidx == {idx}').index[0]`)"""
    return int(np.searchsorted(index.vector_meta.dbidx.values, dbidx, side="left"))


class ActiveSearch(LoopBase):
    def __init__(self, gdm, q, params, weight_matrix: sp.csr_array):
        super().__init__(gdm, q, params)
        self.scores = None
        opts = params.interactive_options
        n = q.index.vectors.shape[0]
        self.gamma = opts["gamma"]
        if self.gamma["mode"] == "clip":
            calibration = self.gamma["calibration"]
            if calibration == "ground_truth":
                self._calibrator = q.get_calibrator()
                assert self._calibrator is not None
            elif calibration == "sigmoid":
                self._calibrator = FixedCalibrator(a=self.gamma["a"], b=self.gamma["b"], sigmoid=True)
            else:
                assert calibration == "raw", f"unknown {calibration=}"
                self._calibrator = FixedCalibrator(a=1.0, b=0.0, sigmoid=False)
            initial_gamma = initial_gamma_array(0.1, n)  # over-written by set_text_vec
        else:
            assert self.gamma["mode"] == "fixed"
            initial_gamma = initial_gamma_array(self.gamma["value"], n)
        self.prob_model = LKNNModel.from_dataset(Dataset.from_vectors(q.index.vectors), gamma=initial_gamma,
                                                 weight_matrix=weight_matrix, device=getattr(q.index, "device", 0) or 0)
        self.dataset = self.prob_model.dataset
        self.pruned_fractions = []
        self.refine_not_called_before = True

    @staticmethod
    def from_params(gdm, q, p):
        return ActiveSearch(gdm, q, p, weight_matrix=get_label_prop(q, p.interactive_options).lp.weight_matrix)

    def set_text_vec(self, tvec):
        super().set_text_vec(tvec)
        self.scores = self.q.index.score(tvec)
        if self.gamma["mode"] == "clip":
            probs = self._calibrator.get_probabilities(tvec, self.q.index.vectors, scores=self.scores)
            self.prob_model = self.prob_model.with_gamma(np.asarray(probs, dtype=np.float64))

    def get_stats(self):
        return {"pruned_fractions": self.pruned_fractions}

    def next_batch(self):
        opts = self.params.interactive_options
        remaining = opts["max_steps"] - len(self.q.returned) if opts["adjust_horizon"] else math.inf
        horizon = int(min(opts["reward_horizon"], remaining))
        assert horizon > 0, "need a non-negative horizon for reward to be defined"
        res = efficient_nonmyopic_search(self.prob_model, reward_horizon=horizon, lookahead_limit=min(2, horizon),
                                         pruning_on=opts["pruning_on"], implementation=opts["implementation"])
        print(f"{res.index=}, {res.value=}")
        self.pruned_fractions.append(res.pruned_fraction)
        abs_idx = self.q.index.vector_meta["dbidx"].iloc[np.array([int(res.index)])].values
        ans = {"dbidxs": abs_idx, "activations": None}
        self.q.returned.update(ans["dbidxs"])
        return ans

    def refine(self, change=None):
        assert change is not None
        print(f"updating model with {change=}")
        if self.refine_not_called_before:  # getXy already includes this latest update
            pos, neg = self.q.getXy(get_positions=True)
            translated = [(int(i), 1) for i in pos] + [(int(i), 0) for i in neg]
        else:
            translated = [(_vector_position(self.q.index, idx), y) for idx, y in change]
        for idx, y in translated:
            self.prob_model.condition_(idx, y)
        self.refine_not_called_before = False


class LKNNSearch(LoopBase):
    def __init__(self, gdm, q, params, weight_matrix: sp.csr_array):
        super().__init__(gdm, q, params)
        opts = params.interactive_options
        self._calibrator = q.get_calibrator()  # debug / experiments only
        if opts["gamma"] == "calibrate":
            assert self._calibrator is not None
            gamma_mean = self._calibrator.get_mean()
        else:
            gamma_mean = opts["gamma"]
        n = q.index.vectors.shape[0]
        self.use_clip_as_gamma = opts["use_clip_as_gamma"]
        self.prob_model = LKNNModel.from_dataset(Dataset.from_vectors(q.index.vectors), gamma=initial_gamma_array(gamma_mean, n),
                                                 weight_matrix=weight_matrix, device=getattr(q.index, "device", 0) or 0)
        self.dataset = self.prob_model.dataset

    @staticmethod
    def from_params(gdm, q, p):
        return LKNNSearch(gdm, q, p, weight_matrix=get_label_prop(q, p.interactive_options).lp.weight_matrix)

    def set_text_vec(self, tvec):
        super().set_text_vec(tvec)
        self.scores = self.q.index.score(tvec)
        if self.use_clip_as_gamma:
            probs = self.scores if self._calibrator is None else self._calibrator.get_probabilities(tvec, self.q.index.vectors, scores=self.scores)
            self.prob_model = self.prob_model.with_gamma(np.asarray(probs, dtype=np.float64))

    def next_batch(self):
        vec_idx, _ = self.prob_model.top_k_remaining(top_k=1)
        print(f"{vec_idx=}")
        abs_idx = self.q.index.vector_meta["dbidx"].iloc[vec_idx].values
        ans = {"dbidxs": abs_idx, "activations": None}
        self.q.returned.update(ans["dbidxs"])
        return ans

    def refine(self, change=None):
        assert change is not None
        print(f"updating model with {change=}")
        for idx, y in change:
            self.prob_model.condition_(_vector_position(self.q.index, idx), y)
