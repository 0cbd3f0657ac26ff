import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from seesaw_amd import _lib
lib = _lib.load()
for rep in range(2):
    for M, N, K, epi, what in [(10000, 3072, 768, 2, "fc1"), (10000, 2304, 768, 1, "qkv"), (20000, 3072, 768, 2, "fc1 x2 rows")]:
        ms, md = ctypes.c_float(), ctypes.c_float()
        lib.ssw_debug_gemm(M, N, K, epi, 9, 20, ctypes.byref(ms), ctypes.byref(md))
        print(os.environ.get("SSW_T256_ABL", "0"), what, f"{ms.value*1e3:.1f} us", flush=True)
