"""PseudoLR: propagate labels over the k-NN graph, then fit the logistic scorer on the real
labels plus a sample of pseudo-labelled vectors (seesaw/loops/pseudo_lr.py:10-54)."""
import numpy as np

from ..logistic_regression import LogisticRegressionPT
from .graph_based import KnnProp2, get_label_prop
from .point_based import PointBased
from .util import makeXy_rows


class PseudoLR(PointBased):
    """two rankers side by side: a KnnProp2 over the k-NN graph produces pseudo-labels (and serves the batches
    until both classes have a real label, when `switch_over` is set); the logistic scorer fitted on real +
    pseudo labels serves them afterwards"""

    def __init__(self, gdm, q, params):
        PointBased.__init__(self, gdm, q, params)
        opts = self.options = params.interactive_options
        for name in ("label_prop_params", "log_reg_params", "switch_over", "real_sample_weight"):
            setattr(self, name, opts[name])
        if not self.real_sample_weight >= 1.0:
            raise AssertionError("real labels must weigh at least as much as pseudo-labels")
        self.knn_based = KnnProp2(gdm, q, params, knn_model=get_label_prop(q, label_prop_params=self.label_prop_params))

    def set_text_vec(self, tvec):
        for loop in (super(), self.knn_based):
            loop.set_text_vec(tvec)

    def refine(self, change=None):
        self.knn_based.refine()
        rows, y, is_real = makeXy_rows(self.knn_based.state.knn_model, sample_size=self.options["sample_size"])
        params = dict(self.log_reg_params)
        params["max_iter"] = int(params.get("max_iter", 100))
        model = LogisticRegressionPT(regularizer_vector=self.state.tvec, device=getattr(self.index, "device", 0), **params)
        weights = np.ones_like(y)
        weights[is_real > 0] = self.real_sample_weight
        dev = getattr(self.index, "_dev", None)
        if dev is not None:
            model.fit(None, y.reshape(-1, 1), weights.reshape(-1, 1), index=dev, rows=rows)
        else:
            model.fit(self.index.vectors[rows], y.reshape(-1, 1), weights.reshape(-1, 1))
        self.curr_vec = model.get_coeff().reshape(-1)

    def next_batch(self):
        pos, neg = self.q.getXy(get_positions=True)
        if self.switch_over and (len(pos) == 0 or len(neg) == 0):
            print("not switching over yet")
            return self.knn_based.next_batch()
        return super().next_batch()
