"""CoarseIndex: one vector per image, exact cosine top-k on the GPU.

Interface of seesaw/indices/coarse/coarse_index.py:16-134 (CoarseIndex, CoarseQuery).  The
reference masks the included rows, copies them (`vectors[metas]`), runs `vecs @ q` and a
full `np.argsort` per query (:67-78); here the matrix stays resident in HBM, the excluded
images are skipped through a device bitmap, and the scan + exact top-k run in
libseesaw_hip.so (ssw_index_topk).
"""
from __future__ import annotations

import json
import os

import numpy as np
import pandas as pd

from ...bitmap import BitMap, FrozenBitMap
from ...device_index import DeviceIndex
from ...query_interface import AccessMethod, InteractiveQuery
from ..interface import ActivationFrames, resolve_path


def _positions_of(sorted_dbidx: np.ndarray, ids) -> np.ndarray:
    ids = np.asarray(list(ids) if not isinstance(ids, np.ndarray) else ids, dtype=np.int64).reshape(-1)
    if ids.size == 0:
        return ids
    pos = np.searchsorted(sorted_dbidx, ids)
    pos = np.clip(pos, 0, sorted_dbidx.shape[0] - 1)
    return pos[sorted_dbidx[pos] == ids]


class CoarseIndex(AccessMethod):
    def __init__(self, embedding, vectors: np.ndarray, vector_meta: pd.DataFrame, path: str = None,
                 device: int = 0):
        self.path = path
        self.embedding = embedding
        self.vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        self.vector_meta = vector_meta
        dbidx = np.asarray(vector_meta.dbidx.values, dtype=np.int64)
        assert np.all(np.diff(dbidx) > 0), "one vector per image, sorted by dbidx (coarse_index.py:49)"
        self._dbidx = dbidx
        self.all_indices = FrozenBitMap(dbidx)
        self.device = device
        self._dev = DeviceIndex.from_numpy(self.vectors, device=device)

    def __len__(self):
        return len(self.all_indices)

    def string2vec(self, string: str) -> np.ndarray:
        vec = self.embedding.from_string(string=string)
        return vec / np.linalg.norm(vec)

    def score(self, tvec) -> np.ndarray:
        return self._dev.scores(tvec)

    @staticmethod
    def from_path(index_path: str, *, use_vec_index=False, exclude=None, device: int = 0, **_):
        """<index>/vectors.npy [N,512] f32 + <index>/vector_meta.parquet (dbidx, ...) + info.json."""
        index_path = resolve_path(index_path)
        info = json.load(open(f"{index_path}/info.json"))
        from ...models.embeddings import load_embedding
        embedding = load_embedding(info.get("model"), device=device)
        vectors = np.load(f"{index_path}/vectors.npy", mmap_mode="r")
        meta = pd.read_parquet(f"{index_path}/vector_meta.parquet")
        assert meta.dbidx.is_monotonic_increasing, "sanity check"
        return CoarseIndex(embedding=embedding, vectors=np.asarray(vectors), vector_meta=meta,
                           path=index_path, device=device)

    def query(self, *, topk, vector=None, exclude=None, startk=None, **kwargs):
        exclude = BitMap() if exclude is None else exclude
        excl_pos = _positions_of(self._dbidx, np.asarray(exclude, dtype=np.int64))
        n_included = self._dbidx.shape[0] - excl_pos.shape[0]
        if n_included == 0:
            return np.array([]), np.array([])
        topk = min(int(topk), n_included)
        if vector is None:  # random order over the included images (coarse_index.py:70-71)
            mask = np.ones(self._dbidx.shape[0], dtype=bool)
            mask[excl_pos] = False
            pos = np.random.permutation(np.nonzero(mask)[0])[:topk]
            scores = np.random.randn(topk)
        else:
            pos, scores, _ = self._dev.topk(vector, topk, excluded=excl_pos)
        ret = self._dbidx[pos]
        assert ret.shape[0] == topk
        # one whole-image box per result (coarse_index.py:88-93), as lazily built one-row frames
        boxes = np.tile(np.array([[0, 0, 224, 224]], dtype=np.int64), (ret.shape[0], 1))
        return {"dbidxs": ret, "nextstartk": len(exclude) + ret.shape[0],
                "activations": ActivationFrames(boxes, ret, np.asarray(scores))}

    def new_query(self):
        return CoarseQuery(self)

    def subset(self, indices: BitMap):
        mask = np.isin(self._dbidx, np.asarray(indices, dtype=np.int64))
        return CoarseIndex(embedding=self.embedding, vectors=self.vectors[mask],
                           vector_meta=self.vector_meta[mask].reset_index(drop=True), device=self.device)


class CoarseQuery(InteractiveQuery):
    def __init__(self, db: CoarseIndex):
        super().__init__(db)

    def getXy(self, get_positions=False):
        seen = np.asarray(self.label_db.get_seen(), dtype=np.int64)
        positions = _positions_of(self.index._dbidx, seen)
        # the reference takes all_indices.rank(idx) - 1 for every seen id (coarse_index.py:116-118), which for
        # an id outside the index silently lands on its predecessor's vector; labels and vectors must pair up
        assert positions.shape[0] == seen.shape[0], "labels recorded for images that are not in this index"
        yt = np.array([len(self.label_db.get(int(d), format="box")) > 0 for d in seen], dtype=bool)
        if get_positions:
            return positions[yt], positions[~yt]
        return self.index.vectors[positions], yt.astype("float")
