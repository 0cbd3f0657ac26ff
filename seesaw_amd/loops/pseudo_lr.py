"""PseudoLR: propagate labels over the k-NN graph, then fit the logistic scorer on the real
labels plus a sample of pseudo-labelled vectors (seesaw/loops/pseudo_lr.py:10-54)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from ..logistic_regression import LogisticRegressionPT
from .graph_based import KnnProp2, get_label_prop
from .point_based import PointBased
from .util import draw_unlabelled, makeXy_rows

_SIDE = None  # one helper thread: the propagation's C call runs there (ctypes drops the GIL) beside the draw


def _overlap_from() -> int:
    """graphs from this many vectors on run the propagation beside the draw (below, the helper thread's hand-over costs
    more than the propagation); SSW_PSEUDOLR_OVERLAP_FROM overrides it (tests force the path on small fixtures)"""
    import os
    return int(os.environ.get("SSW_PSEUDOLR_OVERLAP_FROM", 1 << 18))


def _side():
    global _SIDE
    if _SIDE is None:
        _SIDE = ThreadPoolExecutor(max_workers=1, thread_name_prefix="seesaw-lp")
    return _SIDE


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
        model = self.knn_based.state.knn_model
        lp = getattr(model, "lp", None)
        if lp is not None and hasattr(model, "propagate_now") and model.nvecs >= _overlap_from():
            # KnnProp2.refine with its two halves apart: the labels are recorded, then the propagation (GPU, a few
            # hundred us at 1.56 M vectors) runs on the helper thread while this one draws the pseudo-labelled sample
            # -- which depends on how many vectors are labelled, not on their propagated scores (the reference draws
            # after propagating, pseudo_lr.py:33-35; nothing else touches numpy's generator in between)
            pos, neg = self.q.getXy(get_positions=True)
            model.update_labels(np.concatenate([pos, neg]), np.concatenate([np.ones_like(pos), np.zeros_like(neg)]))
            side = _side().submit(model.propagate_now)
            try:
                drawn = draw_unlabelled(model, self.options["sample_size"], device=lp.device)
            finally:
                side.result()
        else:
            self.knn_based.refine()
            drawn = None
        params = dict(self.log_reg_params)
        params["max_iter"] = int(params.get("max_iter", 100))
        scorer = LogisticRegressionPT(regularizer_vector=self.state.tvec, device=getattr(self.index, "device", 0), **params)
        dev = getattr(self.index, "_dev", None)
        on_device = (dev is not None and lp is not None and getattr(model, "_label_map", None) is not None
                     and getattr(model, "scores_on_device", lambda: False)() and params.get("class_weights") != "balanced")
        if on_device:
            # the training set never exists on the host: labelled rows + labels and the draw go down, the drawn rows'
            # propagated scores become their targets on the device (ssw_fb_set_pseudo_sample) -- makeXy_rows' values, rows
            # and order (tests: the reference's pseudo_lr sessions)
            if drawn is None:
                drawn = draw_unlabelled(model, self.options["sample_size"], device=lp.device)
            lab = model._sorted_label_ids() if hasattr(model, "_sorted_label_ids") else \
                np.fromiter(sorted(model._label_map), dtype=np.int64, count=len(model._label_map))
            scorer.fit(None, None, None, index=dev,
                       pseudo=(lp.device_scores_ptr(), lab, model.labels[lab], drawn, float(self.real_sample_weight)))
        else:
            rows, y, is_real = makeXy_rows(model, sample_size=self.options["sample_size"], drawn=drawn)
            weights = np.ones_like(y)
            weights[is_real > 0] = self.real_sample_weight
            if dev is not None:
                scorer.fit(None, y.reshape(-1, 1), weights.reshape(-1, 1), index=dev, rows=rows)
            else:
                scorer.fit(self.index.vectors[rows], y.reshape(-1, 1), weights.reshape(-1, 1))
        self.curr_vec = scorer.get_coeff().reshape(-1)

    def next_batch(self):
        pos, neg = self.q.getXy(get_positions=True)
        if self.switch_over and (len(pos) == 0 or len(neg) == 0):
            print("not switching over yet")
            return self.knn_based.next_batch()
        return super().next_batch()
