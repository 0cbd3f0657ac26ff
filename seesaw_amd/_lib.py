"""ctypes binding of libseesaw_hip.so (the C-ABI declared in include/seesaw_hip.h).

The HIP library is the product path: if it is missing or fails to load this module
raises -- there is no CPU fallback anywhere in seesaw_amd.
"""
from __future__ import annotations

import ctypes
import os
import re

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSW_PRODUCT_LIB") or os.path.join(_PKG_DIR, "libseesaw_hip.so")  # (the override: A/B of builds)
DEBUG_LIB_PATH = os.environ.get("SSW_DEBUG_LIB") or os.path.join(_PKG_DIR, "libseesaw_hip_debug.so")  # the lab build: same sources + include/seesaw_hip_debug.h
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "seesaw_hip.h")
DEBUG_HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "seesaw_hip_debug.h")

SSW_OK = 0
SSW_ERR_INVALID, SSW_ERR_HIP, SSW_ERR_NOMEM, SSW_ERR_UNSUPPORTED, SSW_ERR_NUMERIC = -1, -2, -3, -4, -5
SSW_MAX_TOPK = 4096


class SeesawHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libseesaw_hip status {status}: {message}")
        self.status = status


c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64
c_void_p = ctypes.c_void_p
c_f32_p = ctypes.POINTER(ctypes.c_float)
c_f64_p = ctypes.POINTER(ctypes.c_double)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_i64_p = ctypes.POINTER(ctypes.c_int64)
c_u64_p = ctypes.POINTER(ctypes.c_uint64)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes).  Must list every symbol include/seesaw_hip.h declares
# (tests/test_abi.py checks the two against each other).
_SIGNATURES = {
    "ssw_abi_version": (c_i32, []),
    "ssw_last_error": (ctypes.c_char_p, []),
    "ssw_device_count": (c_i32, [c_i32_p]),
    "ssw_device_info": (c_i32, [c_i32, ctypes.c_char_p, c_i32, c_i32_p, c_i64_p]),
    "ssw_index_create": (c_i32, [c_i32, c_i64, c_i32, c_void_p, c_void_pp]),
    "ssw_index_destroy": (c_i32, [c_void_p]),
    "ssw_index_set_stream": (c_i32, [c_void_p, c_void_p]),
    "ssw_index_sync": (c_i32, [c_void_p]),
    "ssw_index_shape": (c_i32, [c_void_p, c_i64_p, c_i32_p, c_i64_p]),
    "ssw_index_device_ptrs": (c_i32, [c_void_p, c_void_pp, c_void_pp]),
    "ssw_index_upload": (c_i32, [c_void_p, c_void_p, c_i64, c_i64]),
    "ssw_index_download": (c_i32, [c_void_p, c_void_p, c_i64, c_i64]),
    "ssw_index_fill_random": (c_i32, [c_void_p, c_u64, c_i64]),
    "ssw_index_set_row2image": (c_i32, [c_void_p, c_void_p, c_i64]),
    "ssw_index_scan": (c_i32, [c_void_p, c_void_p, c_void_p]),
    "ssw_index_scan_dev": (c_i32, [c_void_p, c_void_p]),
    "ssw_index_load_scores": (c_i32, [c_void_p, c_void_p]),
    "ssw_index_topk": (c_i32, [c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p,
                               c_void_p, c_i32_p]),
    "ssw_index_set_excluded": (c_i32, [c_void_p, c_void_p, c_i64]),
    "ssw_index_topk_dev": (c_i32, [c_void_p, c_void_p, c_i32]),
    "ssw_index_set_tile_meta": (c_i32, [c_void_p, c_void_p, c_void_p]),
    "ssw_index_rescore_avg": (c_i32, [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "ssw_index_rescore_avg_f64": (c_i32, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p]),
    "ssw_index_select_deep_dev": (c_i32, [c_void_p, c_i32]),
    "ssw_index_result_ptrs": (c_i32, [c_void_p, c_void_pp, c_void_pp, c_void_pp]),
    "ssw_index_topk_fetch": (c_i32, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i32_p]),
    "ssw_index_gather_scores": (c_i32, [c_void_p, c_void_p, c_i64, c_void_p]),
    "ssw_index_score_rows": (c_i32, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "ssw_topk_merge_dev": (c_i32, [c_i32, c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_i32,
                                   c_void_p, c_void_p]),
    "ssw_labelprop_create": (c_i32, [c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_pp]),
    "ssw_labelprop_set_permutation": (c_i32, [c_void_p, c_void_p]),
    "ssw_labelprop_create_ordered": (c_i32, [c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_pp]),
    "ssw_labelprop_destroy": (c_i32, [c_void_p]),
    "ssw_labelprop_run": (c_i32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                                  ctypes.c_double, ctypes.c_double, c_i32, c_void_p, c_i32_p, c_i32_p]),
    "ssw_fb_create": (c_i32, [c_i32, c_i32, c_void_pp]),
    "ssw_fb_destroy": (c_i32, [c_void_p]),
    "ssw_fb_set_data": (c_i32, [c_void_p, c_void_p, c_i64, c_i32]),
    "ssw_fb_set_data_from_device": (c_i32, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_i32]),
    "ssw_fb_set_targets": (c_i32, [c_void_p, c_void_p, c_void_p]),
    "ssw_fb_set_pseudo_sample": (c_i32, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64,
                                         ctypes.c_float, c_i32]),
    "ssw_fb_set_query": (c_i32, [c_void_p, c_void_p]),
    "ssw_labelprop_set_prior": (c_i32, [c_void_p, c_void_p]),
    "ssw_labelprop_run_resident": (c_i32, [c_void_p, c_void_p, c_void_p, c_i64, ctypes.c_double, ctypes.c_double, c_i32, c_void_p, c_void_p]),
    "ssw_labelprop_prior_as_result": (c_i32, [c_void_p, c_void_p, c_i64]),
    "ssw_labelprop_fetch": (c_i32, [c_void_p, c_void_p]),
    "ssw_labelprop_gather": (c_i32, [c_void_p, c_void_p, c_i64, c_void_p]),
    "ssw_labelprop_last_run_info": (c_i32, [c_void_p, c_void_p]),
    "ssw_labelprop_scores_to_index": (c_i32, [c_void_p, c_void_p, c_i32]),
    "ssw_labelprop_device_scores": (c_i32, [c_void_p, ctypes.POINTER(c_void_p)]),
    "ssw_labelprop_round": (c_i32, [c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_i64, ctypes.c_double, ctypes.c_double, c_i32,
                                    c_i32, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssw_xlx": (c_i32, [c_void_p, c_void_p, c_void_p]),
    "ssw_knn_build": (c_i32, [c_void_p, c_i32, ctypes.c_uint64, c_void_p, c_void_p, c_void_p]),
    "ssw_fb_set_xlx": (c_i32, [c_void_p, c_void_p]),
    "ssw_fb_set_targets2": (c_i32, [c_void_p, c_void_p, c_void_p]),
    "ssw_fb_lossgrad2": (c_i32, [c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "ssw_fb_fit2": (c_i32, [c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, c_i32, ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "ssw_fb_get_mean": (c_i32, [c_void_p, c_void_p]),
    "ssw_fb_lossgrad": (c_i32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssw_fb_scores": (c_i32, [c_void_p, c_void_p, c_i32, c_void_p]),
    "ssw_fb_fit": (c_i32, [c_void_p, c_void_p, c_void_p, c_i32, ctypes.c_float, c_i32_p, c_i32_p, c_void_p]),
    "ssw_fb_last_fit_on_device": (c_i32, [c_void_p, c_i32_p]),
    "ssw_fb_reset": (c_i32, [c_void_p]),
    "ssw_np_permutation_prefix": (c_i32, [c_void_p, c_i32_p, c_i64, c_i64, c_void_p]),
    "ssw_np_permutation_prefix_dev": (c_i32, [c_i32, c_void_p, c_i32_p, c_i64, c_i64, c_void_p]),
    "ssw_rank_quick_gradient": (c_i32, [c_i32, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_void_p]),
    "ssw_rank_inversions": (c_i32, [c_i32, c_void_p, c_void_p, c_i32, c_void_p]),
    "ssw_lknn_create": (c_i32, [c_i32, c_i64, c_i32, c_void_p, c_void_pp]),
    "ssw_lknn_destroy": (c_i32, [c_void_p]),
    "ssw_lknn_top_sum": (c_i32, [c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_void_p]),
    "ssw_rank_pairwise": (c_i32, [c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_i32, ctypes.c_float, c_void_p, c_void_p]),
    "ssw_clip_create": (c_i32, [c_i32, c_void_p, ctypes.c_size_t, c_void_pp]),
    "ssw_clip_destroy": (c_i32, [c_void_p]),
    "ssw_clip_embed_image": (c_i32, [c_void_p, c_void_p, c_i32, c_i32, c_void_p]),
    "ssw_clip_embed_image_dev": (c_i32, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_void_p]),
    "ssw_clip_embed_text": (c_i32, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p]),
    "ssw_clip_embed_tiles_u8": (c_i32, [c_void_p, c_void_p, c_i32, c_i32, c_void_p]),
    "ssw_clip_sync": (c_i32, [c_void_p]),
    "ssw_clip_set_option": (c_i32, [c_void_p, c_i32, c_i32]),
    "ssw_wm_build_symmetric": (c_i32, [c_i32, c_i64, c_i64, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_i64)]),
    "ssw_wm_fetch": (c_i32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssw_wm_destroy": (c_i32, [c_void_p]),
    "ssw_index_gather_rows": (c_i32, [c_void_p, c_void_p, c_i64, c_void_p]),
    "ssw_index_set_exchange_target": (c_i32, [c_void_p, c_void_p, c_i32, c_i32, c_i64, c_i64]),
    "ssw_topk_merge_msgs_dev": (c_i32, [c_i32, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ssw_comm_unique_id": (c_i32, [c_void_p]),
    "ssw_comm_create": (c_i32, [c_i32, c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p)]),
    "ssw_comm_destroy": (c_i32, [c_void_p]),
    "ssw_topk_allgather": (c_i32, [c_void_p, c_void_p, c_void_p, c_void_p, c_i32]),
    "ssw_index_profile": (c_i32, [c_void_p, c_i32]),
    "ssw_index_profile_read": (c_i32, [c_void_p, c_void_p, c_i32, c_i32_p]),
}


# include/seesaw_hip_debug.h: the lab build's extra entry points (libseesaw_hip_debug.so only)
_DEBUG_SIGNATURES = {
    "ssw_tune_scan": (c_i32, [c_i32, c_i32]),
    "ssw_tune_topk": (c_i32, [c_i32]),
    "ssw_tune_gemm": (c_i32, [c_i32]),
    "ssw_debug_gemm": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "ssw_debug_gemm_pw4_mode": (c_i32, [c_i32, c_void_p]),
    "ssw_debug_gemm_pw4_wg": (c_i32, [c_void_p]),
    "ssw_debug_gemm_run": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_i32, c_void_p, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]),
    "ssw_debug_attn_out_run": (c_i32, [c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       ctypes.c_float]),
    "ssw_debug_attn_out_stamps": (c_i32, [c_void_p, c_i32]),
    "ssw_clip_debug_tap": (c_i32, [c_void_p, c_i32, c_i32]),
    "ssw_clip_debug_tap_read": (c_i32, [c_void_p, c_void_p, c_i64, c_i64_p, c_i32_p]),
}


class FbObjective(ctypes.Structure):
    """struct ssw_fb_objective"""
    _fields_ = [("kind", c_i32), ("loss_type", c_i32), ("fit_intercept", c_i32), ("reg_kind", c_i32),
                ("pos_weight", ctypes.c_float), ("reg_weight", ctypes.c_float), ("margin", ctypes.c_float),
                ("reg_norm_lambda", ctypes.c_float), ("reg_data_lambda", ctypes.c_float),
                ("reg_query_lambda", ctypes.c_float)]


SSW_FB_LOGREG, SSW_FB_MULTIREG, SSW_FB_RANKREG = 0, 1, 2
SSW_FB_LOSS_CE, SSW_FB_LOSS_PAIRWISE_HINGE, SSW_FB_LOSS_PAIRWISE_LOGISTIC = 0, 1, 2
SSW_FB_REG_NONE, SSW_FB_REG_VECTOR, SSW_FB_REG_NORM, SSW_FB_REG_NORM1 = 0, 1, 2, 3

_lib = None
_product_lib = None
_debug_lib = None


def declared_symbols(header_path: str = HEADER_PATH):
    """Function names declared in include/seesaw_hip.h."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssw_[a-z0-9_]+)\s*\(", text)))


def _open(path: str, signatures):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C seesaw_amd/csrc` (hipcc --offload-arch=gfx950). seesaw_amd has no CPU fallback.")
    # One HIP runtime per process: torch ships its own libamdhip64 and whichever copy is mapped
    # first owns the GPU ("No HIP GPUs are available" from the other).  Loading torch first makes
    # the dynamic linker bind this library's libamdhip64 dependency to torch's copy, so the
    # C-ABI, torch tensors and torch.distributed (RCCL) share one runtime in either import order.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load(path: str = LIB_PATH):
    """dlopen the HIP library (the product build) and install the argument types.  Fails loudly.  Inside a
    `debug_hooks()` block this returns the lab build instead."""
    global _lib, _product_lib
    if _lib is not None:
        return _lib
    _product_lib = _open(path, _SIGNATURES)
    _lib = _product_lib
    return _lib


def load_debug(path: str = DEBUG_LIB_PATH):
    """the lab build (libseesaw_hip_debug.so): everything the product exports plus include/seesaw_hip_debug.h"""
    global _debug_lib
    if _debug_lib is None:
        _debug_lib = _open(path, {**_SIGNATURES, **_DEBUG_SIGNATURES})
    return _debug_lib


class debug_hooks:
    """`with _lib.debug_hooks() as lib:` -- every `_lib.call` / `_lib.load()` inside the block goes to the lab build, so
    kernel variants can be flipped (ssw_tune_*) and single kernels driven (ssw_debug_*).  Handles are plain heap objects
    whose layout is the same in both builds (the only lab-only state, ssw_clip's tap fields, is declared in both and
    freed by either destroy), so a handle may cross the block's boundary; still, create and use handles inside the
    block where you can -- the two libraries hold separate copies of every static (kernel-variant words, attribute
    caches).  Tests and tools only: product code never enters it."""

    def __enter__(self):
        global _lib
        load()
        self._prev = _lib
        _lib = load_debug()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def last_error() -> str:
    msg = load().ssw_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int):
    if status != SSW_OK:
        raise SeesawHipError(status, last_error())


def call(name: str, *args):
    check(getattr(load(), name)(*args))
