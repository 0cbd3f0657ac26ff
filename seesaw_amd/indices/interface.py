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
