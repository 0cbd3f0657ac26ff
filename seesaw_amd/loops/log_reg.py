"""LogReg2: refit a regularised logistic scorer on the labels after every batch and query
with its coefficient vector (seesaw/loops/log_reg.py:5-33)."""
from ..logistic_regression import LogisticRegressionPT
from .point_based import PointBased


class LogReg2(PointBased):
    def __init__(self, gdm, q, params):
        super().__init__(gdm, q, params)
        self.model = None

    @staticmethod
    def from_params(gdm, q, params):
        return LogReg2(gdm, q, params)

    def set_text_vec(self, vec):
        super().set_text_vec(vec)
        self.model = None

    def refine(self, change=None):
        xy = self.q.getXy()
        if isinstance(xy, tuple):
            Xt, yt = xy
        else:  # multiscale query: labelled tile rows
            Xt, yt = self.q.index.vectors[xy.index.values], xy.ys.values
        if self.model is None:
            self.model = LogisticRegressionPT(regularizer_vector=self.state.tvec,
                                              device=getattr(self.q.index, "device", 0),
                                              **self.params.interactive_options)
        if (yt == 1).all():
            print("doing nothing, only positives")
        elif (yt == 0).all():
            print("doing nothing, only negatives")
        else:
            self.model.fit(Xt, yt.reshape(-1, 1))
            self.curr_vec = self.model.get_coeff()
