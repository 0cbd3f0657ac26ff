"""AccessMethod: the abstract index the session and the loops talk to.

Contract of seesaw/indices/interface.py:10-45: the attribute and method names, the keyword-only
signatures and the `info.json["constructor"]` dispatch of `load` are the reference's.  A concrete index
(CoarseIndex, MultiscaleIndex) keeps `.vectors` on the host for the callers that read it directly
(multi_reg.py:204, rocchio_update.py:24, loops/util.py:6) and a `DeviceIndex` with the same matrix in HBM
for every scan; `.vector_meta` has one row per vector (dbidx, and for tiles zoom_level / x1 y1 x2 y2).
"""
from __future__ import annotations

import json
import os

import numpy as np

from ..basic_types import get_constructor
from ..bitmap import BitMap


def resolve_path(path: str) -> str:
    """absolute, symlink-free, `~`-expanded"""
    return os.path.normpath(os.path.realpath(os.path.expanduser(path)))


def _not_here(what: str):
    def stub(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not implement {what}")
    stub.__name__ = what
    return stub


class ActivationFrames:
    """the `activations` entry of a query result: one one-row DataFrame [x1, y1, x2, y2, dbidx, score] per returned
    image, as the reference builds them (multiscale_index.py:389-392) -- materialised only when somebody indexes or
    iterates.  A pandas frame costs ~150 us to construct; the session layer, the one consumer on the interactive
    path, only wants the numbers (`records()`), so a batch-size-1 round no longer builds a frame at all."""

    _COLS = ["x1", "y1", "x2", "y2", "dbidx", "score"]

    def __init__(self, boxes, dbidx, scores):
        self._boxes, self._dbidx, self._scores = np.asarray(boxes), np.asarray(dbidx), np.asarray(scores)
        self._frames = {}

    def __len__(self):
        return int(self._dbidx.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        f = self._frames.get(i)
        if f is None:
            import pandas as pd
            b = self._boxes[i]
            f = self._frames[i] = pd.DataFrame({"x1": [b[0]], "y1": [b[1]], "x2": [b[2]], "y2": [b[3]],
                                                "dbidx": [self._dbidx[i]], "score": [self._scores[i]]})
        return f

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def records(self):
        """[(x1, y1, x2, y2, score)] as python floats"""
        b = self._boxes.astype(np.float64)
        s = self._scores.astype(np.float64)
        return [(float(b[i, 0]), float(b[i, 1]), float(b[i, 2]), float(b[i, 3]), float(s[i])) for i in range(len(self))]


class AccessMethod:
    path: str = None

    # text -> unit query vector [1, d]
    string2vec = _not_here("string2vec")
    # query(*, vector, topk, exclude=None, **kwargs) -> {"dbidxs": int array, "activations": list | None}
    query = _not_here("query")
    # score(vec) -> [N] f32, one score per vector
    score = _not_here("score")
    # new_query() -> InteractiveQuery bound to this index
    new_query = _not_here("new_query")
    # subset(BitMap of dbidxs) -> AccessMethod over those images only
    subset = _not_here("subset")

    def get_knng_path(self, name: str = None) -> str:
        """directory of a named k-NN graph of this index (`forward.parquet` inside)"""
        return f"{self.path}/knn_graph/{name or ''}"

    @staticmethod
    def from_path(index_path: str, **options):
        raise NotImplementedError("concrete indices construct themselves from their directory")

    @staticmethod
    def load(index_path: str, *, options: dict = None, exclude=None):
        """open an index directory: `info.json` names the constructor (a `seesaw.` class path is mapped onto
        this package), whose `from_path` does the rest"""
        root = resolve_path(index_path)
        with open(os.path.join(root, "info.json")) as f:
            constructor = get_constructor(json.load(f)["constructor"])
        return constructor.from_path(root, **(options or {}), exclude=exclude)
