"""CLIP text tower, batched path (16 x 77 tokens: the tile kernels), for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seesaw_amd.models.clip import ClipModel

B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), (int(sys.argv[2]) if len(sys.argv) > 2 else 77)
m = ClipModel.random_init(seed=1234)
ids = np.random.default_rng(0).integers(0, 49405, (B, L)).astype(np.int32)
ids[:, 0], ids[:, -1] = 49406, 49407
for _ in range(3):
    m.embed_text(ids)
t0 = time.perf_counter()
for _ in range(20):
    m.embed_text(ids)
dt = (time.perf_counter() - t0) / 20
print(f"text {B} x {L}: {dt*1e3:.3f} ms host to host, {B * 5.96 * (L / 77.0) / dt / 1e3:.1f} TFLOP/s", flush=True)
