"""LogReg2 (seesaw/loops/log_reg.py:5-33): after every batch a regularised logistic scorer is refitted on all
labels so far (GPU L-BFGS through LogisticRegressionPT) and its coefficient vector becomes the query."""
import numpy as np

from ..logistic_regression import LogisticRegressionPT
from .point_based import PointBased


def _labelled_xy(q):
    """(X, y) of the labelled vectors for either kind of query object: CoarseQuery.getXy returns the pair,
    MultiscaleQuery.getXy a frame indexed by vector position"""
    got = q.getXy()
    if isinstance(got, tuple):
        return got
    return q.index.vectors[got.index.values], got.ys.values


class LogReg2(PointBased):
    model = None  # built lazily on the first refine, dropped when the text changes

    def set_text_vec(self, vec):
        PointBased.set_text_vec(self, vec)
        self.model = None

    def refine(self, change=None):
        X, y = _labelled_xy(self.q)
        y = np.asarray(y)
        if self.model is None:
            self.model = LogisticRegressionPT(regularizer_vector=self.state.tvec, device=getattr(self.q.index, "device", 0),
                                              **self.params.interactive_options)
        one_sided = {1: "positives", 0: "negatives"}
        for value, what in one_sided.items():
            if (y == value).all():
                print(f"doing nothing, only {what}")
                return
        self.model.fit(X, y.reshape(-1, 1))
        self.curr_vec = self.model.get_coeff()
