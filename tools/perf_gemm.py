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
    """hipBLASLt through torch on the same shapes: the plain C = A W^T (bf16 out, no epilogue), and the same
    product followed by the elementwise work the fused kernels do in their epilogue, as a torch program would
    run it (bias through the library's own epilogue, then quick-GELU / the f32 residual add as separate kernels)."""
    import torch

    dev = torch.device("cuda", 0)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20

    for M, N, K, epi, what in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
        bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
        res = torch.randn(M, N, device=dev, dtype=torch.float32)
        lin = torch.nn.functional.linear
        plain = timed(lambda: lin(a, w))
        if epi == 0:
            full = plain
        elif epi == 1:
            full = timed(lambda: lin(a, w, bias))
        elif epi == 2:
            def f():
                h = lin(a, w, bias)
                return h * torch.sigmoid(1.702 * h)
            full = timed(f)
        else:
            full = timed(lambda: res + lin(a, w, bias).float())
        flop = 2.0 * M * N * K
        print(f"{what:10s} hipBLASLt(torch) plain {plain*1e3:7.1f}us {flop/(plain*1e-3)/1e12:6.0f}TF | with the epilogue's work "
              f"{full*1e3:7.1f}us {flop/(full*1e-3)/1e12:6.0f}TF", flush=True)


if __name__ == "__main__" and "--lib" in sys.argv:
    library_baseline()
