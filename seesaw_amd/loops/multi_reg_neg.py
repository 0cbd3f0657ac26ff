"""MultiRegNeg: the query vector plus a second vector for the class the user says the results are being confused
with; batches are ranked by score(target) - score(confusion) when `discount_neg` is set
(seesaw/loops/multi_reg_neg.py:26-109)."""
import numpy as np

from ..feedback import FeedbackEngine
from .multi_reg_module import MultiRegModule
from .point_based import PointBased


class MultiRegNeg(PointBased):
    def __init__(self, gdm, q, params):
        super().__init__(gdm, q, params)
        from .graph_based import get_weight_matrix_from_index
        self.options = self.params.interactive_options
        # the reference builds X'LX here and never uses it (its data regulariser is commented out,
        # multi_reg_module.py:115); built only when asked for so a missing graph is reported the same way
        self.xlx_matrix = None
        if self.options.get("matrix_options") is not None:
            self.xlx_matrix = get_weight_matrix_from_index(q.index, self.options["matrix_options"], xlx_matrix=True)
        self.confusion_vec = None
        self._engine = FeedbackEngine(q.index.vectors.shape[1], device=getattr(q.index, "device", 0))

    def set_text_vec(self, tvec):
        super().set_text_vec(tvec)
        if self.options["reg_data_lambda"] > 0 and self.options["reg_query_lambda"] > 0 and self.started:
            self.refine()
        else:
            self.curr_vec = self.curr_qvec

    def refine(self, change=None):
        matchdf = self.q.getXy(target_description=None)
        rows = matchdf.index.values
        y = matchdf.ys.values
        box_df = self.q.label_db.get_box_df(return_description=True)
        descs = box_df[box_df.marked_accepted == 0].description.unique()
        if len(descs) > 0:  # the first description the user rejected names the confusion class
            alt_desc = descs[0]
            print(f"{alt_desc=}")
            yconf = self.q.getXy(target_description=alt_desc).ys.values
        else:
            yconf = np.zeros_like(y)
        ys = np.stack([y, yconf], axis=1).astype("float32")
        assert ys.shape[0] == y.shape[0] and ys.shape[1] == 2
        print(f"{ys.sum(axis=0)=}")
        assert self.curr_qvec is not None
        o = self.options
        model = MultiRegModule(qvec=self.curr_qvec, reg_norm_lambda=o["reg_norm_lambda"],
                               reg_query_lambda=o["reg_query_lambda"], verbose=o["verbose"], max_iter=int(o["max_iter"]),
                               lr=o["lr"], dim=self.q.index.vectors.shape[1], engine=self._engine)
        dev = getattr(self.q.index, "_dev", None)
        if dev is not None:  # labelled rows are gathered out of the resident index, no upload
            model.fit(None, ys, matchdf, index=dev, rows=rows)
        else:
            model.fit(self.q.index.vectors[rows], ys, matchdf)
        self.curr_vec = model.get_coeff()
        self.confusion_vec = model.get_confusion_coeff()

    def next_batch(self):
        dim = self.q.index.vectors.shape[1]
        return self.q.query_stateful(
            vector=self.curr_vec, batch_size=self.params.batch_size, shortlist_size=self.params.shortlist_size,
            agg_method=self.params.agg_method, aug_larger=self.params.aug_larger, rescore_method=None,
            vector2=self.confusion_vec if self.options["discount_neg"] else np.zeros(dim))
