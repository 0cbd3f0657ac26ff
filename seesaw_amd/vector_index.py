"""VectorIndex: `query(vector, top_k) -> (ids, scores)` over a vector file.

Same surface as seesaw/vector_index.py:9-60.  The reference wraps an annoy ("dot", 512-d)
approximate index; here the exact brute-force scan on the GPU replaces it (results are a
superset in quality of what annoy returns; parity is against `_get_top_exact`).  The file
`load_path` points at is a plain `[N, 512]` float32 matrix (`.npy`).
"""
from __future__ import annotations

import time

import numpy as np

from .device_index import DeviceIndex


def build_annoy_idx(*, vecs, output_path, n_trees=None):
    """Write the vector file VectorIndex loads (name kept from the reference, where it builds
    an annoy forest: vector_index.py:9-19).  There is no build step for an exact scan, so this
    only materialises the matrix; returns the elapsed seconds like the reference."""
    start = time.time()
    with open(output_path, "wb") as f:  # explicit handle: np.save would append ".npy"
        np.save(f, np.ascontiguousarray(vecs, dtype=np.float32))
    return time.time() - start


def build_nndescent_idx(vecs, output_path, n_trees=None):
    """kept for signature parity (vector_index.py:22-41); same file as build_annoy_idx."""
    return build_annoy_idx(vecs=vecs, output_path=output_path, n_trees=n_trees)


class VectorIndex:
    def __init__(self, *, load_path=None, prefault=False, vectors: np.ndarray = None, device: int = 0):
        if vectors is None:
            vectors = np.load(load_path, mmap_mode=None if prefault else "r")
        assert vectors.ndim == 2 and vectors.shape[1] == 512, "VectorIndex holds [N, 512] vectors"
        self._dev = DeviceIndex.from_numpy(np.asarray(vectors), device=device)
        self.n = vectors.shape[0]

    def ready(self):
        return True

    def query(self, vector, top_k):
        assert vector.shape == (1, 512) or vector.shape == (512,)
        ids, scores, _ = self._dev.topk(vector, min(int(top_k), self.n))
        return ids, scores
