"""one tower-GEMM shape on one variant (for rocprofv3): python3 tools/perf_gemm_one.py M N K epi variant [iters]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from seesaw_amd import _lib

_lib.debug_hooks().__enter__()  # the lab build (libseesaw_hip_debug.so): ssw_tune_* / ssw_debug_* live there

M, N, K, epi, variant = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
lib = _lib.load()
ms, md = ctypes.c_float(), ctypes.c_float()
if lib.ssw_debug_gemm(M, N, K, epi, variant, iters, ctypes.byref(ms), ctypes.byref(md)) != 0:
    raise RuntimeError(lib.ssw_last_error().decode())
print(f"M={M} N={N} K={K} epi={epi} v{variant}: {ms.value*1e3:.1f} us {2.0*M*N*K/(ms.value*1e-3)/1e12:.0f} TF d={md.value:.2e}")
