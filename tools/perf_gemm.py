"""A/B the tower GEMM variants in one process (GPU box): ms, TFLOP/s and max |diff| to variant 0."""
import ctypes

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: its bundled HIP runtime must be the one the process uses)

from seesaw_amd import _lib

_lib.debug_hooks().__enter__()  # the lab build (libseesaw_hip_debug.so): ssw_tune_* / ssw_debug_* live there

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


def pw_diag():
    """where a K-step of the persistent kernel spends its cycles (fc1: 256 x 256 tiles; patch: 256 x 128), and what the
    loop costs without its LDS-DMA / MFMAs / fragment reads"""
    import numpy as np
    lib = _lib.load()
    for (M, N, K, epi, what, v) in [(10000, 3072, 768, 2, "fc1+gelu", 21), (9800, 768, 3072, 0, "patch", 23)]:
        line = f"{what:10s} pw4:"
        for mode in (0, 1, 2, 3, 4, 5):
            lib.ssw_debug_gemm_pw4_mode(mode, None)
            ms, md = ctypes.c_float(), ctypes.c_float()
            out = np.zeros(6, dtype=np.uint64)
            lib.ssw_debug_gemm_pw4_mode(mode, ctypes.c_void_p(out.ctypes.data))  # reset the counters
            rc = lib.ssw_debug_gemm(M, N, K, epi, v, 20, ctypes.byref(ms), ctypes.byref(md))
            if rc != 0:
                raise RuntimeError(lib.ssw_last_error().decode())
            line += f" | mode {mode} {ms.value*1e3:6.1f}us"
            if mode == 1:
                lib.ssw_debug_gemm_pw4_mode(1, ctypes.c_void_p(out.ctypes.data))
                wait, steps, n, waves, ticks, rticks = (float(x) for x in out)
                buf = np.zeros(4096 + 20, dtype=np.uint64)
                lib.ssw_debug_gemm_pw4_wg(ctypes.c_void_p(buf.ctypes.data))
                wg = buf[:4096].reshape(1024, 4)
                wg = wg[wg[:, 1] > 0].astype(np.int64)
                t0 = wg[:, 0].min()
                dur = (wg[:, 1] - wg[:, 0]) / 100.0
                start = (wg[:, 0] - t0) / 100.0
                cu = (wg[:, 2] >> 8) & 0xF, 
                ident = wg[:, 3] * 100000 + ((wg[:, 2] >> 13) & 0x7) * 1000 + ((wg[:, 2] >> 12) & 1) * 100 + ((wg[:, 2] >> 8) & 0xF)
                print(f"   workgroups {wg.shape[0]}: lifetime us min/med/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f}; start offsets us "
                      f"min/med/max {start.min():.1f}/{np.median(start):.1f}/{start.max():.1f}; distinct (xcc, se, sh, cu) ids {np.unique(ident).shape[0]}; "
                      f"span {((wg[:, 1].max() - t0) / 100.0):.1f} us")
                line += (f" (per K-step {steps / max(n, 1):.0f} cycles, of them wait+barrier {wait / max(n, 1):.0f}; "
                         f"s_memtime runs at {ticks / max(rticks, 1) * 100:.0f} MHz)")
            if mode == 5:
                buf = np.zeros(4096 + 20, dtype=np.uint64)
                lib.ssw_debug_gemm_pw4_wg(ctypes.c_void_p(buf.ctypes.data))
                grp = buf[4096:].astype(np.float64)
                nb = max(grp[17], 1.0)
                print("   mode 5, cycles per interleave group (2 reads | piece, 4 MFMAs): " + " ".join(f"{v / nb:.0f}" for v in grp[:16])
                      + f" ; between bodies (waits + barrier) {grp[16] / nb:.0f}")
        lib.ssw_debug_gemm_pw4_mode(0, None)
        print(line, flush=True)


if __name__ == "__main__" and "--pw-diag" in sys.argv:
    pw_diag()
elif __name__ == "__main__":
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
