"""Interleaved A/B of the scan kernel's schedule variants in ONE process (development aid;
guides/cdna_hip_programming.md rule 24): python tools/sweep_scan.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seesaw_amd import _lib
_lib.debug_hooks().__enter__()  # the lab build (libseesaw_hip_debug.so): ssw_tune_* / ssw_debug_* live there
from seesaw_amd.device_index import DeviceIndex

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 32_000_000
idx = DeviceIndex.synthetic(n, 512, seed=1)
q = np.random.default_rng(0).standard_normal(512).astype(np.float32)
q /= np.linalg.norm(q)
names = {0: "u4", 1: "u4nt", 2: "u8", 3: "u8nt", 4: "u2nt", -1: "dflt"}
configs = [(v, b) for v in (0, 1, 3, 4) for b in (0, 4, 2, 1)] + [(-1, -1)]
res = {c: [] for c in configs}
ref = None
for rnd in range(4):
    for c in configs:
        _lib.call("ssw_tune_scan", c[0], c[1])
        idx.scan(q)
        idx.profile(True)
        for _ in range(6):
            idx.scan(q)
        ms = idx.profile_read()
        idx.profile(False)
        res[c].append(float(np.median(ms)))
        s = idx.topk(None, 10)[1]
        if ref is None:
            ref = s
        assert np.array_equal(ref.view(np.uint32), s.view(np.uint32)), "variants must be bit-identical"
for c in configs:
    ms = np.array(res[c])
    print(f"{names[c[0]]:5s} blocks/CU cap {c[1]}: median {np.median(ms):.3f} ms  min {ms.min():.3f}  "
          f"-> {n*2048/np.median(ms)/1e6:.0f} GB/s (best {n*2048/ms.min()/1e6:.0f})", flush=True)
