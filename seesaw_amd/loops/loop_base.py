"""LoopBase: what every feedback loop shares (interface of seesaw/loops/loop_base.py:17-106):
the current query vector, the start policy gate, and `next_batch_external` /
`refine_external`, which the session calls."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..basic_types import SessionParams
from ..query_interface import InteractiveQuery


@dataclass
class LoopState:
    curr_str: str = None
    tvec: np.ndarray = None
    vec_state = None
    knn_model = None


class LoopBase:
    q: InteractiveQuery
    params: SessionParams
    state: LoopState

    def __init__(self, gdm, q: InteractiveQuery, params: SessionParams):
        self.gdm = gdm
        self.params = params
        self.state = LoopState()
        self.q = q
        self.index = q.index
        self.curr_qvec = None
        self.reversal = False  # set by the session
        self.started = params.start_policy == "from_start"

    def set_reversals(self):
        if not self.reversal:
            print("first reversal seen...")
            self.reversal = True

    def get_stats(self):
        return None

    def set_text_vec(self, vec):
        self.curr_qvec = vec

    def _next_batch_curr_vec(self, vec):
        assert not np.isnan(vec).any(), f"NaN in query vector {vec=}"
        p = self.params
        return self.q.query_stateful(vector=vec, batch_size=p.batch_size, shortlist_size=p.shortlist_size,
                                     agg_method=p.agg_method, aug_larger=p.aug_larger,
                                     rescore_method=lambda vecs: vecs @ vec.reshape(-1, 1))

    @classmethod
    def from_params(cls, gdm, q, params) -> "LoopBase":
        """the registry's constructor hook (every loop of the reference defines the same three-argument
        static method; here the concrete loops inherit it)"""
        if cls is LoopBase:
            raise NotImplementedError
        return cls(gdm, q, params)

    def next_batch_external(self):
        if self.started:
            print("start met. next batch from custom method...")
            return self.next_batch()
        print("start not yet met. next batch using default...")
        return self._next_batch_curr_vec(vec=self.curr_qvec)

    def next_batch(self):
        raise NotImplementedError("implement me in subclass")

    def refine(self, change=None):
        raise NotImplementedError("implement me in subclass")

    def _start_condition(self) -> bool:
        policy = self.params.start_policy
        if policy == "from_start":
            return True
        if policy == "after_first_reversal":
            return self.reversal
        fast = getattr(self.q, "_matched_arrays", None)
        if callable(fast):  # per-image max of ys without building the frame + groupby (1 ms of a session's first round)
            rows, miou = fast(None)
            ys = miou > 0
            dbidx = self.q.index._row_dbidx[rows]
            pos_images = np.unique(dbidx[ys])
            len_pos, len_neg = int(pos_images.shape[0]), int(np.setdiff1d(np.unique(dbidx), pos_images).shape[0])
            xy = None
        else:
            xy = self.q.getXy()
        if xy is None:
            pass
        elif isinstance(xy, tuple):  # coarse query: (X, y)
            ys = np.asarray(xy[1])
            len_pos, len_neg = int((ys == 1).sum()), int((ys == 0).sum())
        else:
            by_image = xy.groupby("dbidx").ys.max()
            len_pos, len_neg = int((by_image == 1.0).sum()), int((by_image == 0.0).sum())
        if policy == "after_first_batch":
            return (len_pos + len_neg) > 0
        if policy == "after_first_positive":
            return len_pos > 0
        if policy == "after_first_negative":
            return len_neg > 0
        if policy == "after_first_positive_and_negative":
            return len_pos > 0 and len_neg > 0
        raise AssertionError("policy not implemented")

    def refine_external(self, change=None):
        if not self.started:
            self.started = self._start_condition()
        if self.started:
            print("start condition met... refinining custom method...")
            self.refine(change=change)
