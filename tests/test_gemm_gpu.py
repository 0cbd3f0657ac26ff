"""GPU: the towers' tile GEMMs (csrc/gemm_bf16.hip) on ragged row counts.

`ssw_debug_gemm` runs one kernel variant on seeded operands and returns its largest difference to variant 0, the
register-staged kernel with the plain per-lane epilogue.  The LDS-DMA kernels store through LDS in whole row segments
with the row tail masked at the store (round 3), so the cases here are row counts around the 128- and 256-row tile
edges, on every epilogue and on both default kernels.  Tolerance: the variants add a row's products in a different
order than variant 0 -- one bf16 ulp at |x| < 4 for the bf16 outputs (1.5625e-2), f32 rounding for the f32 ones; a
wrong row, column or mask is an O(1) difference."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(768, 768), (2304, 768), (768, 3072), (512, 2048)]  # (N, K): attn-out, QKV, fc2, text fc2


@pytest.mark.parametrize("variant", [15, 9, 14])
@pytest.mark.parametrize("M", [1, 17, 127, 129, 255, 257, 650, 1000])
def test_tile_gemm_ragged_rows_against_the_register_staged_kernel(M, variant):
    import torch  # noqa: F401  (first: its bundled HIP runtime must be the one the process uses)
    from seesaw_amd import _lib
    lib = _lib.load()
    for N, K in SHAPES:
        for epi in (0, 1, 2, 3):
            ms, md = ctypes.c_float(), ctypes.c_float()
            rc = lib.ssw_debug_gemm(M, N, K, epi, variant, 1, ctypes.byref(ms), ctypes.byref(md))
            assert rc == 0, lib.ssw_last_error().decode()
            tol = 3.2e-2 if epi in (1, 2) else 2e-4
            assert md.value <= tol, (M, N, K, epi, variant, md.value)
