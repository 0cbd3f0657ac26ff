"""Image tower, 200 tiles: one call of 200 against two concurrent calls of 100 on two streams (two handles with the same
weights, each with its own workspace) -- does a second kernel chain fill the first one's launch tails?  (GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from seesaw_amd.models.clip import ClipModel

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ms = [ClipModel.random_init(seed=1234) for _ in range(4)]
x = torch.randn(B, 3, 224, 224, device=dev)
out = torch.empty(B, 512, device=dev)
ref = torch.empty(B, 512, device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]
torch.cuda.synchronize()


def run(parts):
    step = B // parts
    for p in range(parts):
        lo = p * step
        hi = B if p == parts - 1 else lo + step
        ms[p].embed_image_dev(x[lo:hi].data_ptr(), hi - lo, (ref if parts == 1 else out)[lo:hi].data_ptr(), True,
                              streams[p].cuda_stream)


for parts in (1, 2, 4, 1, 2):
    for _ in range(3):
        run(parts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run(parts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    same = bool((out == ref).all().item()) if parts > 1 else True
    print(f"B={B} in {parts} concurrent call(s): {dt * 1e3:.3f} ms, {B * 8.818 / dt / 1e3:.1f} TFLOP/s, identical vectors: {same}",
          flush=True)
