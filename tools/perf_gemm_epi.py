"""The tile GEMMs with the epilogues the image tower runs them with (4 LN -> QKV, 5 LN -> fc1 + GELU, 6 f32 rows +
statistics, 7 bf16 rows + statistics) beside their plain forms (1, 2, 3), each alone and back to back: what the
folded LayerNorm and the residual-row traffic cost a launch (lab build; GPU box)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from seesaw_amd import _lib

_lib.debug_hooks().__enter__()
lib = _lib.load()
CASES = [("qkv", 10000, 2304, 768, (1, 4)), ("fc1", 10000, 3072, 768, (2, 5)), ("attn-out", 10000, 768, 768, (3, 6, 7)),
         ("fc2", 10000, 768, 3072, (3, 6, 7))]
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 14
for what, M, N, K, epis in CASES:
    line = f"{what:9s} M={M} N={N} K={K}"
    for epi in epis:
        ms, md = ctypes.c_float(), ctypes.c_float()
        rc = lib.ssw_debug_gemm(M, N, K, epi, variant, 20, ctypes.byref(ms), ctypes.byref(md))
        if rc != 0:
            raise RuntimeError(lib.ssw_last_error().decode())
        line += f" | epi {epi} {ms.value * 1e3:6.1f} us {2.0 * M * N * K / (ms.value * 1e-3) / 1e12:5.0f} TF"
    print(line, flush=True)
