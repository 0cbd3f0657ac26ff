"""A/B the tower GEMM variants in one process (GPU box): ms, TFLOP/s and max |diff| to variant 0."""
import ctypes

import torch  # noqa: F401  (first: its bundled HIP runtime must be the one the process uses)
import sys

from seesaw_amd import _lib

SHAPES = [  # (M, N, K, epi, what)
    (10000, 2304, 768, 1, "qkv"),
    (10000, 768, 768, 3, "attn-out"),
    (10000, 3072, 768, 2, "fc1+gelu"),
    (10000, 768, 3072, 3, "fc2+res"),
    (9800, 768, 3072, 0, "patch"),
    (15400, 1536, 512, 1, "text qkv"),
    (15400, 2048, 512, 2, "text fc1"),
    (15400, 512, 2048, 3, "text fc2"),
]


def main():
    lib = _lib.load()
    variants = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [0, 2, 14, 7]
    iters = 20
    for M, N, K, epi, what in SHAPES:
        line = f"{what:10s} M={M:6d} N={N:5d} K={K:5d} epi={epi}"
        for v in variants:
            ms, md = ctypes.c_float(), ctypes.c_float()
            rc = lib.ssw_debug_gemm(M, N, K, epi, v, iters, ctypes.byref(ms), ctypes.byref(md))
            if rc != 0:
                raise RuntimeError(lib.ssw_last_error().decode())
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
            line += f" | v{v} {ms.value*1e3:7.1f}us {tf:6.0f}TF d={md.value:.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()


def library_baseline():
    """hipBLASLt through torch (plain C = A W^T, bf16 out, no epilogue) on the same shapes."""
    import torch

    dev = torch.device("cuda", 0)
    for M, N, K, epi, what in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
        for _ in range(3):
            torch.nn.functional.linear(a, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.nn.functional.linear(a, w)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{what:10s} hipBLASLt(torch) {ms*1e3:7.1f}us {2.0*M*N*K/(ms*1e-3)/1e12:6.0f}TF", flush=True)


if __name__ == "__main__" and "--lib" in sys.argv:
    library_baseline()
