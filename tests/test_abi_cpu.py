"""CPU: the C-ABI library loads and exports every symbol include/seesaw_hip.h declares, the
ctypes binding covers exactly that set, and the host-side helpers behave (no GPU compute)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from seesaw_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "seesaw_amd", "csrc"), "-j", "4"], check=True)
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = lib.declared_symbols()
    assert len(declared) >= 30
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in seesaw_hip.h but not exported"
    assert sorted(lib._SIGNATURES) == declared, "ctypes table and header disagree"
    nm = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ssw_[a-z0-9_]+)", nm))
    assert exported == set(declared), (exported ^ set(declared))


def test_lab_build_holds_the_hooks_and_the_product_does_not(lib):
    """ssw_tune_* / ssw_debug_* live in include/seesaw_hip_debug.h and exist only in
    libseesaw_hip_debug.so (compiled with -DSSW_DEBUG_HOOKS); the product header declares none of them and the product
    library exports none of them -- its kernel-selection switches are constants."""
    product = set(lib.declared_symbols())
    hooks = set(lib.declared_symbols(lib.DEBUG_HEADER_PATH))
    assert hooks and not (hooks & product)
    assert not [n for n in product if n.startswith(("ssw_tune_", "ssw_debug_"))]
    assert sorted(lib._DEBUG_SIGNATURES) == sorted(hooks), "ctypes table and debug header disagree"
    nm = subprocess.run(["nm", "-D", "--defined-only", lib.DEBUG_LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ssw_[a-z0-9_]+)", nm))
    assert exported == product | hooks, (exported ^ (product | hooks))
    dbg = lib.load_debug()
    for name in product | hooks:
        assert hasattr(dbg, name)
    # no mutable tuning word in the product: the switches are `const` there (SSW_TUNABLE), so they are not data symbols
    nm_all = subprocess.run(["nm", "-C", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for word in ("g_gemm_variant", "g_scan_variant", "g_select_sampled", "g_small_path", "g_clip_flags"):
        assert not re.search(r" [BbDd] .*" + word, nm_all), word
    with lib.debug_hooks() as inside:
        assert inside is dbg and lib.load() is dbg
    assert lib.load() is not dbg


def test_abi_version_and_error_channel(lib):
    h = lib.load()
    assert h.ssw_abi_version() == 1
    # an invalid call reports through the status + ssw_last_error, without touching a GPU
    out = ctypes.c_void_p()
    st = h.ssw_index_create(0, 10, 500, None, ctypes.byref(out))  # dim not a multiple of 256
    assert st == -4 and "dim=500" in lib.last_error()
    st = h.ssw_fb_create(0, 513, ctypes.byref(out))
    assert st == -4


def test_objective_struct_layout_matches_header(lib):
    text = open(lib.HEADER_PATH).read()
    body = re.search(r"typedef struct ssw_fb_objective \{(.*?)\} ssw_fb_objective;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int32_t|float)\s+([a-z_]+);", body)
    assert [f for _, f in fields] == [f for f, _ in lib.FbObjective._fields_]
    assert ctypes.sizeof(lib.FbObjective) == 4 * len(fields)


def test_missing_library_fails_loudly(lib, tmp_path):
    import importlib
    from seesaw_amd import _lib as fresh
    saved = fresh._lib
    fresh._lib = None
    try:
        with pytest.raises(ImportError, match="no CPU fallback"):
            fresh.load(str(tmp_path / "libseesaw_hip.so"))
    finally:
        fresh._lib = saved


def test_bitmap_semantics():
    from seesaw_amd.bitmap import BitMap, FrozenBitMap
    a = BitMap([5, 1, 9, 5])
    assert list(a) == [1, 5, 9] and len(a) == 3 and 5 in a and 4 not in a
    assert np.array_equal(np.array(a), [1, 5, 9])
    assert a.rank(5) == 2 and a.rank(0) == 0 and a.rank(100) == 3
    a.update([7, 1])
    a.add(2)
    assert list(a) == [1, 2, 5, 7, 9]
    f = FrozenBitMap([2, 9, 11])
    assert list(a - f) == [1, 5, 7] and list(a.intersection(f)) == [2, 9] and a.intersection_cardinality(f) == 2
    assert list(f.union([0])) == [0, 2, 9, 11] and isinstance(f - a, FrozenBitMap)
    b = a.copy()
    b.clear()
    assert len(b) == 0 and len(a) == 5
    assert f.intersection(a) == FrozenBitMap([2, 9])


def test_shard_bounds():
    from seesaw_amd.sharded import shard_bounds, shard_bounds_by_image
    for n, w in [(100_000_000, 8), (10, 3), (7, 8), (0, 2)]:
        edges = [shard_bounds(n, w, r) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        assert max(hi - lo for lo, hi in edges) - min(hi - lo for lo, hi in edges) <= 1
    tiles = np.random.default_rng(0).integers(1, 30, 1000)
    row_start = np.concatenate(([0], np.cumsum(tiles)))
    parts = [shard_bounds_by_image(row_start, 4, r) for r in range(4)]
    assert parts[0][0] == 0 and parts[-1][1] == 1000 and parts[-1][3] == row_start[-1]
    for a, b in zip(parts, parts[1:]):
        assert a[1] == b[0] and a[3] == b[2]
    sizes = [p[3] - p[2] for p in parts]
    assert max(sizes) - min(sizes) <= 2 * tiles.max()


def test_metrics_known_answers():
    # values of the reference's tests/test_metrics.py (signatures there are stale, values hold)
    from seesaw_amd.metrics import average_precision, ndcg_score, rank_of_kth
    assert average_precision(np.array([0, 1, 2]), npositive=3) == 1.0
    assert average_precision(np.array([0, 1, 2]), npositive=4, max_results=3) == 1.0
    assert abs(average_precision(np.array([1, 3]), npositive=2) - (1 / 2 + 2 / 4) / 2) < 1e-12
    assert average_precision(np.array([]), npositive=3) == 0.0
    assert ndcg_score(np.array([0, 1]), nseen=2, npositive=2) == 1.0
    assert rank_of_kth(np.array([4, 7]), ntotal=10, k=2) == 8
    assert rank_of_kth(np.array([4]), ntotal=10, k=2) == float("inf")
    assert rank_of_kth(np.array([4]), ntotal=1, k=2) is None
