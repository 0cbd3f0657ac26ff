"""Latency of one ssw_index_topk call on an LVIS-subset-size index (1 109 images x 13 tiles, dim 512): the fixed cost of
a `plain` feedback round.  Prints us per call for the small-index form and for the general path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seesaw_amd import _lib  # noqa: E402
_lib.debug_hooks().__enter__()  # the lab build (libseesaw_hip_debug.so): ssw_tune_* / ssw_debug_* live there
from seesaw_amd.device_index import DeviceIndex  # noqa: E402


def main():
    n_images, tiles = 1109, 13
    r2i = np.repeat(np.arange(n_images), tiles).astype(np.int32)
    idx = DeviceIndex.synthetic(n_images * tiles, 512, seed=3, row2image=r2i) if "row2image" in DeviceIndex.synthetic.__code__.co_varnames \
        else DeviceIndex.from_numpy(np.random.default_rng(0).standard_normal((n_images * tiles, 512)).astype(np.float32), row2image=r2i)
    q = np.random.default_rng(1).standard_normal(512).astype(np.float32)
    for mode, name in ((1, "small-index form"), (0, "general path")):
        _lib.call("ssw_tune_topk", mode)
        for reps in (200, 2000):
            ex = []
            t0 = time.perf_counter()
            for i in range(reps):
                imgs, _, _ = idx.topk(q, 60, excluded=ex)
                if i % 67 == 0:
                    ex = list(range((i // 67) * 30 % 900))
            dt = time.perf_counter() - t0
        print(f"{name}: {1e6 * dt / reps:.1f} us per ssw_index_topk (k=60, <= 900 excluded), python wrapper included")
    import ctypes
    imgs, scs, rows = np.empty(60, np.int64), np.empty(60, np.float32), np.empty(60, np.int64)
    cnt = ctypes.c_int32(0)
    ex = np.arange(600, dtype=np.int64)
    fn = _lib.lib().ssw_index_topk if hasattr(_lib, "lib") else None
    for mode, name, variant in ((1, "small-index form", -1), (1, "small-index form, scan u8", 3), (1, "small-index form, scan u4", 1),
                                (0, "general path", -1)):
        _lib.call("ssw_tune_topk", mode)
        _lib.call("ssw_tune_scan", variant, -1)
        for with_q in (True, False):
            args = (idx._h, q.ctypes.data if with_q else None, ex.ctypes.data, 600, 60, imgs.ctypes.data, scs.ctypes.data,
                    rows.ctypes.data, ctypes.byref(cnt))
            for reps in (200, 3000):
                t0 = time.perf_counter()
                for _ in range(reps):
                    _lib.call("ssw_index_topk", *args)
                dt = time.perf_counter() - t0
            print(f"{name}, C call only, {'scan + select' if with_q else 'select only'}: {1e6 * dt / reps:.1f} us")
    _lib.call("ssw_tune_topk", 3)
    idx.close()


if __name__ == "__main__":
    main()
