"""AccessMethod: the abstract index the session / loops talk to.

Mirrors seesaw/indices/interface.py:10-45 (same attribute and method names, same
keyword-only signatures, same `info.json["constructor"]` dispatch in `load`).  Concrete
indices keep `.vectors` (numpy, host) for the callers that read it directly
(multi_reg.py:204, rocchio_update.py:24, loops/util.py:6) and a `DeviceIndex` holding the
same matrix in HBM for every scan.
"""
from __future__ import annotations

import json
import os

import numpy as np

from ..basic_types import get_constructor
from ..bitmap import BitMap


def resolve_path(path: str) -> str:
    return os.path.normpath(os.path.realpath(os.path.expanduser(path)))


class AccessMethod:
    path: str = None

    def string2vec(self, string: str) -> np.ndarray:
        raise NotImplementedError("implement me")

    def query(self, *, vector: np.ndarray, topk: int, exclude: BitMap = None, **kwargs) -> dict:
        raise NotImplementedError("implement me")

    def score(self, vec: np.ndarray) -> np.ndarray:
        raise NotImplementedError("implement me")

    def new_query(self):
        raise NotImplementedError("implement me")

    def subset(self, indices: BitMap):
        raise NotImplementedError("implement me")

    def get_knng_path(self, name: str = None) -> str:
        return f"{self.path}/knn_graph/{name or ''}"

    @staticmethod
    def from_path(index_path: str, **options):
        raise NotImplementedError("implement me")

    @staticmethod
    def load(index_path: str, *, options: dict = None, exclude=None):
        index_path = resolve_path(index_path)
        with open(f"{index_path}/info.json") as f:
            meta = json.load(f)
        cls = get_constructor(meta["constructor"])
        return cls.from_path(index_path, **(options or {}), exclude=exclude)
