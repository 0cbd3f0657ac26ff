"""Latency of the selection alone (ssw_index_topk with q = NULL: exclusion delta, per-image maximum, histograms, candidate
pass, final sort, packed result) on an index of LVIS scale: 120 000 images x 13 tiles.   python tools/perf_select_mid.py [n_images]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seesaw_amd import _lib  # noqa: E402
from seesaw_amd.device_index import DeviceIndex  # noqa: E402

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
tiles = 13
idx = DeviceIndex.synthetic(n_images * tiles, 512, seed=3)
idx.set_row2image(np.repeat(np.arange(n_images), tiles).astype(np.int32))
q = np.random.default_rng(1).standard_normal(512).astype(np.float32)
q /= np.linalg.norm(q)
k = 50
imgs, scs, rows = np.empty(k, np.int64), np.empty(k, np.float32), np.empty(k, np.int64)
cnt = ctypes.c_int32(0)
ex = np.arange(0, 600, 20, dtype=np.int64)
base = idx.topk(q, k, excluded=ex.tolist())
for with_q in (True, False):
    args = (idx._h, q.ctypes.data if with_q else None, ex.ctypes.data, ex.shape[0], k, imgs.ctypes.data, scs.ctypes.data,
            rows.ctypes.data, ctypes.byref(cnt))
    for reps in (100, 1000):
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.call("ssw_index_topk", *args)
        dt = time.perf_counter() - t0
    assert np.array_equal(imgs[:cnt.value], base[0]) and np.array_equal(rows[:cnt.value], base[2])
    print(f"{n_images} images: {'scan + select' if with_q else 'select only'} {1e6 * dt / reps:.1f} us per ssw_index_topk")
idx.close()
