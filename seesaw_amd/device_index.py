"""DeviceIndex: the [N, d] f32 vector matrix resident in HBM + scan / top-k over the C-ABI.

This is the object the reference-shaped indices (`seesaw_amd.indices.*`,
`seesaw_amd.vector_index.VectorIndex`) delegate their numeric work to.  It replaces
`vectors @ q` + `np.argsort` + `_get_top_dbidxs` of the reference
(seesaw/indices/multiscale/multiscale_index.py:170-199, coarse_index.py:57-96).
numpy arrays in, numpy arrays out; everything in between runs in libseesaw_hip.so.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, Optional

import numpy as np

from . import _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class DeviceIndex:
    def __init__(self, n_rows: int, dim: int = 512, device: int = 0, dev_ptr: int = 0):
        self._h = ctypes.c_void_p()
        self.n_rows = int(n_rows)
        self.dim = int(dim)
        self.device = int(device)
        self.n_images = self.n_rows
        _lib.call("ssw_index_create", self.device, self.n_rows, self.dim,
                  ctypes.c_void_p(dev_ptr) if dev_ptr else None, ctypes.byref(self._h))

    # -- construction -----------------------------------------------------------------
    @classmethod
    def from_numpy(cls, vectors: np.ndarray, row2image: Optional[np.ndarray] = None,
                   device: int = 0, chunk_rows: int = 1 << 18) -> "DeviceIndex":
        vectors = np.asarray(vectors)
        assert vectors.ndim == 2, "vectors must be [N, d]"
        idx = cls(vectors.shape[0], vectors.shape[1], device=device)
        for r0 in range(0, vectors.shape[0], chunk_rows):
            chunk = np.ascontiguousarray(vectors[r0:r0 + chunk_rows], dtype=np.float32)
            _lib.call("ssw_index_upload", idx._h, _ptr(chunk), r0, chunk.shape[0])
        if row2image is not None:
            idx.set_row2image(row2image)
        return idx

    @classmethod
    def synthetic(cls, n_rows: int, dim: int = 512, seed: int = 0, first_row: int = 0,
                  device: int = 0) -> "DeviceIndex":
        idx = cls(n_rows, dim, device=device)
        _lib.call("ssw_index_fill_random", idx._h, ctypes.c_uint64(seed), int(first_row))
        return idx

    def set_row2image(self, row2image: Optional[np.ndarray]):
        """row2image[r] = position (0..n_images-1) of row r's image; non-decreasing."""
        if row2image is None:
            _lib.call("ssw_index_set_row2image", self._h, None, 0)
            self.n_images = self.n_rows
            return
        r2i = np.ascontiguousarray(row2image, dtype=np.int32)
        assert r2i.shape == (self.n_rows,)
        n_images = int(r2i[-1]) + 1 if self.n_rows else 0
        _lib.call("ssw_index_set_row2image", self._h, _ptr(r2i), n_images)
        self.n_images = n_images

    def close(self):
        if self._h:
            _lib.load().ssw_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data movement ----------------------------------------------------------------
    def download(self, first_row: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = self.n_rows - first_row if n is None else n
        out = np.empty((n, self.dim), dtype=np.float32)
        _lib.call("ssw_index_download", self._h, _ptr(out), int(first_row), int(n))
        return out

    def device_ptrs(self):
        v, s = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("ssw_index_device_ptrs", self._h, ctypes.byref(v), ctypes.byref(s))
        return v.value, s.value

    HIP_STREAM_LEGACY = 1  # hipStreamLegacy: the handle that names the default (NULL) stream explicitly

    def set_stream(self, stream_ptr: int):
        """run this handle's work on the caller's stream.  torch's default stream has the handle 0, which the
        C-ABI reads as "back to the handle's own stream": it is passed as hipStreamLegacy instead, so the scan
        and the selection are ordered with the torch ops issued around them."""
        _lib.call("ssw_index_set_stream", self._h, ctypes.c_void_p(int(stream_ptr) or self.HIP_STREAM_LEGACY))

    def restore_own_stream(self):
        _lib.call("ssw_index_set_stream", self._h, None)

    def sync(self):
        _lib.call("ssw_index_sync", self._h)

    # -- scan / top-k -----------------------------------------------------------------
    def _query(self, q: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(np.asarray(q).reshape(-1), dtype=np.float32)
        assert q.shape[0] == self.dim, f"query has {q.shape[0]} components, index dim is {self.dim}"
        return q

    def scores(self, q: np.ndarray) -> np.ndarray:
        """index.score(vec): all N cosine scores (multiscale_index.py:284-285)."""
        q = self._query(q)
        out = np.empty(self.n_rows, dtype=np.float32)
        _lib.call("ssw_index_scan", self._h, _ptr(q), _ptr(out))
        return out

    def scan(self, q: np.ndarray):
        """Run the scan and leave the scores resident on the device."""
        q = self._query(q)
        _lib.call("ssw_index_scan", self._h, _ptr(q), None)

    def topk(self, q: Optional[np.ndarray], k: int, excluded: Optional[Iterable[int]] = None):
        """Top-k distinct images (positions), their max score and the row attaining it.
        q=None reuses the scores of the previous scan."""
        k = int(k)
        qa = None if q is None else self._query(q)
        ex = None
        n_ex = 0
        if excluded is not None:
            ex = np.ascontiguousarray(np.fromiter(excluded, dtype=np.int64))
            n_ex = ex.shape[0]
            if n_ex == 0:
                ex = None
        imgs = np.empty(k, dtype=np.int64)
        scs = np.empty(k, dtype=np.float32)
        rows = np.empty(k, dtype=np.int64)
        cnt = ctypes.c_int32(0)
        _lib.call("ssw_index_topk", self._h, _ptr(qa), _ptr(ex), n_ex, k, _ptr(imgs), _ptr(scs),
                  _ptr(rows), ctypes.byref(cnt))
        c = cnt.value
        return imgs[:c], scs[:c], rows[:c]

    def load_scores(self, scores: np.ndarray):
        """overwrite the resident per-row scores (f32; -inf rows are never selected)."""
        s = np.ascontiguousarray(scores, dtype=np.float32)
        assert s.shape == (self.n_rows,)
        _lib.call("ssw_index_load_scores", self._h, _ptr(s))

    def gather_scores(self, rows: np.ndarray) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty(rows.shape[0], dtype=np.float32)
        _lib.call("ssw_index_gather_scores", self._h, _ptr(rows), rows.shape[0], _ptr(out))
        return out

    def gather_rows(self, rows: np.ndarray) -> np.ndarray:
        """`vectors[rows]` [n, dim] f32 out of the resident matrix"""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((rows.shape[0], self.dim), dtype=np.float32)
        _lib.call("ssw_index_gather_rows", self._h, _ptr(rows), rows.shape[0], _ptr(out))
        return out

    def score_rows(self, q: np.ndarray, rows: np.ndarray) -> np.ndarray:
        """`vectors[rows] @ q` on the device, same summation order as the scan."""
        q = self._query(q)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty(rows.shape[0], dtype=np.float32)
        _lib.call("ssw_index_score_rows", self._h, _ptr(q), _ptr(rows), rows.shape[0], _ptr(out))
        return out

    # -- second stage: avg_score aggregation ------------------------------------------
    AUG_LARGER = {"all": 0, "greater": 1, "adjacent": 2}

    def set_tile_meta(self, boxes: np.ndarray, zoom_level: np.ndarray):
        """tile boxes [n_rows, 4] = x1, y1, x2, y2 (f32) and zoom levels [n_rows] of every row"""
        b = np.ascontiguousarray(boxes, dtype=np.float32)
        z = np.ascontiguousarray(zoom_level, dtype=np.int32)
        assert b.shape == (self.n_rows, 4) and z.shape == (self.n_rows,)
        _lib.call("ssw_index_set_tile_meta", self._h, _ptr(b), _ptr(z))

    def rescore_avg(self, image_positions: np.ndarray, aug_larger: str, minus_scores: Optional[np.ndarray] = None,
                    aug_weight: str = "level_max"):
        """score_frame2's `avg_score` for the given candidate images over the resident tile scores:
        -> (aggregated score of the image's best tile f32 [m], that tile's row int64 [m])"""
        pos = np.ascontiguousarray(image_positions, dtype=np.int64)
        m = pos.shape[0]
        minus = None if minus_scores is None else np.ascontiguousarray(minus_scores, dtype=np.float32)
        scores = np.empty(m, dtype=np.float32)
        rows = np.empty(m, dtype=np.int64)
        aug = self.AUG_LARGER[aug_larger] | {"level_max": 0, "cont_weighted": 4}[aug_weight]
        _lib.call("ssw_index_rescore_avg", self._h, _ptr(pos), m, aug, _ptr(minus), _ptr(scores), _ptr(rows))
        return scores, rows

    def rescore_avg_f64(self, dev_scores_ptr: int, image_positions: np.ndarray, aug_larger: str,
                        aug_weight: str = "level_max"):
        """rescore_avg over float64 tile scores resident on the device (one per index row, e.g. the label-propagation
        output): -> (aggregated score f64 [m], best tile's row int64 [m])"""
        pos = np.ascontiguousarray(image_positions, dtype=np.int64)
        m = pos.shape[0]
        scores = np.empty(m, dtype=np.float64)
        rows = np.empty(m, dtype=np.int64)
        aug = self.AUG_LARGER[aug_larger] | {"level_max": 0, "cont_weighted": 4}[aug_weight]
        _lib.call("ssw_index_rescore_avg_f64", self._h, ctypes.c_void_p(int(dev_scores_ptr)), _ptr(pos), m, aug,
                  _ptr(scores), _ptr(rows))
        return scores, rows

    # -- device-resident forms (bench / sharded index) --------------------------------
    def set_excluded(self, excluded: Optional[Iterable[int]]):
        ex = None
        n_ex = 0
        if excluded is not None:
            ex = np.ascontiguousarray(np.fromiter(excluded, dtype=np.int64))
            n_ex = ex.shape[0]
            if n_ex == 0:
                ex = None
        _lib.call("ssw_index_set_excluded", self._h, _ptr(ex), n_ex)

    def topk_dev(self, q_dev_ptr: int, k: int):
        _lib.call("ssw_index_topk_dev", self._h, ctypes.c_void_p(q_dev_ptr) if q_dev_ptr else None, int(k))

    def select_deep_dev(self, k: int):
        """exact selection for the mass-tie case (result_ptrs' overflow word set): same result buffers"""
        _lib.call("ssw_index_select_deep_dev", self._h, int(k))

    def scan_dev(self, q_dev_ptr: int):
        _lib.call("ssw_index_scan_dev", self._h, ctypes.c_void_p(q_dev_ptr))

    def result_ptrs(self):
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("ssw_index_result_ptrs", self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def topk_fetch(self, k: int):
        k = int(k)
        imgs = np.empty(k, dtype=np.int64)
        scs = np.empty(k, dtype=np.float32)
        rows = np.empty(k, dtype=np.int64)
        cnt = ctypes.c_int32(0)
        _lib.call("ssw_index_topk_fetch", self._h, k, _ptr(imgs), _ptr(scs), _ptr(rows), ctypes.byref(cnt))
        c = cnt.value
        return imgs[:c], scs[:c], rows[:c]

    # -- k-NN graph -------------------------------------------------------------------
    def knn(self, k: int, seed: int = 0):
        """Exact k nearest rows of every row (compute_exact_knn, knn_graph.py:170-191):
        returns (dst int32 [n, k+1], score f32 [n, k+1], n_recomputed).  Each row lists the k+1
        best rows including itself by (score desc, row id asc); scores are scan-order f32.
        Rows the fp16 candidate pass could not certify are redone with the ordinary exact scan."""
        k = int(k)
        n = self.n_rows
        dst = np.empty((n, k + 1), dtype=np.int32)
        score = np.empty((n, k + 1), dtype=np.float32)
        cert = np.empty(n, dtype=np.uint8)
        _lib.call("ssw_knn_build", self._h, k, ctypes.c_uint64(int(seed)), _ptr(dst), _ptr(score), _ptr(cert))
        redo = np.nonzero(cert == 0)[0]
        if redo.shape[0]:
            view = DeviceIndex(n, self.dim, device=self.device, dev_ptr=self.device_ptrs()[0])  # rows, no image map
            try:
                for r in redo.tolist():
                    ids, sc, _ = view.topk(self.download(r, 1)[0], k + 1)
                    dst[r, :ids.shape[0]] = ids
                    score[r, :ids.shape[0]] = sc
            finally:
                view.close()
        return dst, score, int(redo.shape[0])

    # -- profiling --------------------------------------------------------------------
    def profile(self, enable: bool):
        _lib.call("ssw_index_profile", self._h, int(bool(enable)))

    def profile_read(self) -> np.ndarray:
        out = np.empty(4096, dtype=np.float32)
        n = ctypes.c_int32(0)
        _lib.call("ssw_index_profile_read", self._h, _ptr(out), 4096, ctypes.byref(n))
        return out[:n.value].copy()


def decode_keys(keys: np.ndarray):
    """(score_key << 32 | ~image) composite keys -> (images int64, scores f32)."""
    keys = np.asarray(keys, dtype=np.uint64)
    imgs = (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
    o = (keys >> np.uint64(32)).astype(np.uint32)
    neg = (o & np.uint32(0x80000000)) == 0
    u = np.where(neg, ~o, o & np.uint32(0x7FFFFFFF)).astype(np.uint32)
    return imgs, u.view(np.float32)
