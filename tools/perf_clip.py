"""CLIP tower throughput probe (development aid): python tools/perf_clip.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from seesaw_amd.models.clip import ClipModel

IMG_GFLOP = 8.818   # per 224x224 tile
TXT_GFLOP_77 = 5.96


def main():
    t0 = time.time()
    m = ClipModel.random_init(seed=1234)
    print(f"model ready in {time.time()-t0:.1f}s", flush=True)
    dev = torch.device("cuda", 0)
    for B in (13, 200, 800):
        x = torch.randn(B, 3, 224, 224, device=dev)
        out = torch.empty(B, 512, device=dev)
        torch.cuda.synchronize()
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            m.embed_image_dev(x.data_ptr(), B, out.data_ptr(), True, s)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            m.embed_image_dev(x.data_ptr(), B, out.data_ptr(), True, s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"image B={B}: {dt*1e3:.2f} ms/batch, {B/dt:.0f} tiles/s, {B*IMG_GFLOP/dt/1e3:.1f} TFLOP/s", flush=True)
    for (B, L) in ((1, 8), (16, 77), (256, 77)):
        ids = np.random.default_rng(0).integers(0, 49405, (B, L)).astype(np.int32)
        ids[:, 0] = 49406
        ids[:, -1] = 49407
        m.embed_text(ids)
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            m.embed_text(ids)
        dt = (time.perf_counter() - t0) / n
        print(f"text B={B} L={L}: {dt*1e3:.2f} ms/batch (host in/out), {B/dt:.0f} texts/s", flush=True)


if __name__ == "__main__":
    main()
