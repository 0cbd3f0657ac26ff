"""Exact k-NN graph build timing on device-generated rows (GPU box)."""
import sys
import time

from seesaw_amd.device_index import DeviceIndex


def main():
    sizes = [int(float(a)) for a in sys.argv[1:]] or [200_000, 1_560_000]
    for n in sizes:
        dev = DeviceIndex.synthetic(n, 512, seed=5)
        t0 = time.perf_counter()
        dst, score, redone = dev.knn(10)
        dt = time.perf_counter() - t0
        flop = 2.0 * n * n * 512
        print(f"n={n}: {dt:.3f} s  {flop/dt/1e12:.0f} TFLOP/s (fp16 candidate pass incl. everything)  "
              f"recomputed rows {redone} ({100.0*redone/n:.3f} %)  self-first {float((dst[:,0]==range(n)).mean()):.4f}", flush=True)
        dev.close()


if __name__ == "__main__":
    main()
