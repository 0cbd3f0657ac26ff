"""Rocchio relevance feedback: q = alpha q0 + beta mean(relevant) - gamma mean(non-relevant)
(seesaw/loops/rocchio_update.py:4-39; Manning, Raghavan & Schuetze, IR book p.182)."""
import numpy as np

from .point_based import PointBased


class RocchioUpdate(PointBased):
    def __init__(self, gdm, q, params):
        super().__init__(gdm, q, params)
        o = params.interactive_options
        self.alpha, self.beta, self.gamma = o["rocchio_alpha"], o["rocchio_beta"], o["rocchio_gamma"]

    def refine(self, change=None):
        matchdf = self.q.getXy()
        X = self.q.index.vectors[matchdf.index.values]
        y = matchdf.ys.values
        rel, nrel = X[y > 0], X[y == 0]
        mean_rel = rel.sum(axis=0) / (rel.shape[0] or 1.0)
        mean_nrel = nrel.sum(axis=0) / (nrel.shape[0] or 1.0)
        self.curr_vec = self.alpha * self.curr_qvec + self.beta * mean_rel - self.gamma * mean_nrel
