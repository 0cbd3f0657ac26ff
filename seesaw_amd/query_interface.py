"""InteractiveQuery: per-session wrapper that remembers what was already returned.

Same surface as seesaw/query_interface.py:7-52: `index`, `returned`, `label_db`,
`query_stateful(batch_size=..., **kw)` (pops batch_size -> topk, passes
`exclude=self.returned`, then adds the new dbidxs to `returned`) and the abstract `getXy`.
"""
from __future__ import annotations

import numpy as np

from .bitmap import BitMap
from .indices.interface import AccessMethod
from .labeldb import LabelDB

__all__ = ["InteractiveQuery", "AccessMethod"]


class InteractiveQuery:
    index: AccessMethod
    label_db: LabelDB

    def __init__(self, index: AccessMethod, _y: np.ndarray = None):
        self.index = index
        self.returned = BitMap()   # images handed out by the index (not necessarily labelled yet)
        self.label_db = LabelDB()  # labels that came back
        self._calibrator = None
        if _y is not None:
            from .calibration import GroundTruthCalibrator
            self._calibrator = GroundTruthCalibrator(self.index.vectors, _y)

    def get_calibrator(self):
        return self._calibrator

    def query_stateful(self, *args, **kwargs):
        topk = kwargs.pop("batch_size")
        res = self.index.query(*args, topk=topk, **kwargs, exclude=self.returned)
        self.returned.update(np.asarray(res["dbidxs"], dtype=np.int64))
        return res

    def getXy(self, **options):
        raise NotImplementedError("abstract")
