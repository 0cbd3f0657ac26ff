"""Quick scan-kernel bandwidth probe (development aid): python tools/perf_scan.py [rows ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seesaw_amd.device_index import DeviceIndex


def main():
    sizes = [int(float(a)) for a in sys.argv[1:]] or [1_000_000, 8_000_000]
    for n in sizes:
        idx = DeviceIndex.synthetic(n, 512, seed=1)
        q = np.random.default_rng(0).standard_normal(512).astype(np.float32)
        q /= np.linalg.norm(q)
        idx.scan(q)
        idx.profile(True)
        t0 = time.perf_counter()
        for _ in range(20):
            idx.scan(q)
        wall = (time.perf_counter() - t0) / 20
        ms = idx.profile_read()
        idx.profile(False)
        gbs = n * 2048 / (ms * 1e-3) / 1e9
        print(f"n={n}: scan kernel {np.median(ms):.4f} ms median (min {ms.min():.4f}) -> "
              f"{np.median(gbs):.0f} GB/s median, {gbs.max():.0f} best; host wall/scan {wall*1e3:.3f} ms", flush=True)
        idx.topk(q, 100)  # first call allocates the select workspace
        t0 = time.perf_counter()
        for _ in range(20):
            idx.topk(q, 100)
        wall = (time.perf_counter() - t0) / 20
        print(f"   topk(k=100) host wall {wall*1e3:.3f} ms -> {n/wall/1e9:.3f} G vectors/s", flush=True)
        idx.close()


if __name__ == "__main__":
    main()
