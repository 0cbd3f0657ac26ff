"""GPU parity tests for the scan + top-k path (through the C-ABI) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def DeviceIndex():
    from seesaw_amd.device_index import DeviceIndex
    return DeviceIndex


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 127, 1000, 10007])
def test_synth_rows_bit_exact(DeviceIndex, oracle, n):
    idx = DeviceIndex.synthetic(n, 512, seed=11, first_row=5)
    got = idx.download()
    ref = oracle.synth_rows(11, 5, n, 512)
    assert np.array_equal(bits(got), bits(ref))
    idx.close()


@pytest.mark.parametrize("dim", [256, 512, 768, 1024])
@pytest.mark.parametrize("n", [1, 64, 65, 127, 4097, 20011, 65535, 70001])
def test_scores_bit_exact_vs_kernel_order_oracle(DeviceIndex, oracle, n, dim):
    X = oracle.synth_rows(3, 100, n, dim)
    q = oracle.synth_query(1, dim)
    idx = DeviceIndex.from_numpy(X)
    got = idx.scores(q)
    ref = oracle.scores_kernel_order(X, q)
    assert np.array_equal(bits(got), bits(ref)), f"max abs diff {np.abs(got - ref).max()}"
    # and within f32 rounding of the reference expression `vectors @ q`
    assert np.abs(got - oracle.scores_reference(X, q)).max() <= oracle.rounding_band(X, q)
    idx.close()


@pytest.mark.parametrize("n", [5, 129, 14417, 40000])
def test_small_index_scan_kernel_equals_streaming_kernel(DeviceIndex, oracle, lab_build, n):
    """under 65 536 rows the scan is the latency-shaped kernel; ssw_tune_scan(-2) forces the streaming one: same bits"""
    from seesaw_amd import _lib
    X = oracle.synth_rows(8, 3, n, 512)
    idx = DeviceIndex.from_numpy(X)
    try:
        for qs in range(3):
            q = oracle.synth_query(qs)
            _lib.call("ssw_tune_scan", -2, -1)
            want = idx.scores(q)
            _lib.call("ssw_tune_scan", -1, -1)
            assert np.array_equal(bits(idx.scores(q)), bits(want))
            assert np.array_equal(bits(want), bits(oracle.scores_kernel_order(X, q)))
    finally:
        _lib.call("ssw_tune_scan", -1, -1)
        idx.close()


def test_scores_general_values(DeviceIndex, oracle):
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((5000, 512)) * np.exp(rng.uniform(-20, 20, (5000, 1)))).astype(np.float32)
    q = rng.standard_normal(512).astype(np.float32)
    idx = DeviceIndex.from_numpy(X)
    assert np.array_equal(bits(idx.scores(q)), bits(oracle.scores_kernel_order(X, q)))
    idx.close()


@pytest.mark.parametrize("k", [1, 10, 100, 1000, 4096])
@pytest.mark.parametrize("tiles", [1, 13])
def test_topk_images_vs_oracle(DeviceIndex, oracle, k, tiles):
    n = 50000
    X = oracle.synth_rows(5, 0, n, 512)
    q = oracle.synth_query(2)
    row2image = None if tiles == 1 else (np.arange(n) // tiles).astype(np.int32)
    n_images = n if tiles == 1 else int(row2image[-1]) + 1
    rng = np.random.default_rng(k)
    excluded = rng.choice(n_images, size=min(500, n_images // 2), replace=False)
    idx = DeviceIndex.from_numpy(X, row2image=row2image)
    imgs, scores, rows = idx.topk(q, k, excluded=excluded)
    ref_scores = oracle.scores_kernel_order(X, q)
    o_imgs, o_scores, o_rows = oracle.topk_images_tiebreak(ref_scores, row2image, n_images, excluded, k)
    assert np.array_equal(imgs, o_imgs)
    assert np.array_equal(bits(scores), bits(o_scores))
    assert np.array_equal(rows, o_rows)
    # set parity against the reference expression (BLAS order + argsort), band rule
    ok, msg = oracle.check_topk_against_reference(imgs, oracle.scores_reference(X, q), row2image,
                                                  excluded, k, oracle.rounding_band(X, q))
    assert ok, msg
    idx.close()


def test_topk_ragged_images_and_reuse(DeviceIndex, oracle):
    rng = np.random.default_rng(1)
    counts = rng.integers(1, 40, size=3000)
    row2image = np.repeat(np.arange(3000), counts).astype(np.int32)
    n = row2image.shape[0]
    X = oracle.synth_rows(9, 0, n, 512)
    idx = DeviceIndex.from_numpy(X, row2image=row2image)
    returned = []
    q = oracle.synth_query(4)
    ref_scores = oracle.scores_kernel_order(X, q)
    for rnd in range(5):  # the exclusion list grows like InteractiveQuery.returned
        imgs, scores, rows = idx.topk(q if rnd == 0 else None, 50, excluded=returned)
        o_imgs, o_scores, o_rows = oracle.topk_images_tiebreak(ref_scores, row2image, 3000, returned, 50)
        assert np.array_equal(imgs, o_imgs) and np.array_equal(rows, o_rows)
        assert np.array_equal(bits(scores), bits(o_scores))
        g = idx.gather_scores(rows)
        assert np.array_equal(bits(g), bits(scores))
        returned.extend(imgs[:10].tolist())
    idx.close()


@pytest.mark.parametrize("tiles", [1, 13])
def test_small_index_form_equals_general_path(DeviceIndex, oracle, lab_build, tiles):
    """an index of <= 8192 images runs ssw_index_topk as two launches with pinned, device-mapped query / ids / result;
    switched off (ssw_tune_topk(0)) the same calls take the general path: identical answers, round after round, with the
    list growing, repeated ids, q=None reuse, k beyond what is left, and everything excluded"""
    from seesaw_amd import _lib
    n_images = 1109
    rng = np.random.default_rng(tiles)
    counts = np.full(n_images, tiles) if tiles == 1 else rng.integers(1, 2 * tiles, size=n_images)
    row2image = np.repeat(np.arange(n_images), counts).astype(np.int32)
    X = oracle.synth_rows(21, 0, row2image.shape[0], 512)
    ref_idx = DeviceIndex.from_numpy(X, row2image=None if tiles == 1 else row2image)
    idx = DeviceIndex.from_numpy(X, row2image=None if tiles == 1 else row2image)
    returned = []
    try:
        for rnd in range(12):
            q = oracle.synth_query(rnd // 2)
            k = (10, 60, 4096)[rnd % 3]
            _lib.call("ssw_tune_topk", 0)
            want = ref_idx.topk(q if rnd % 2 == 0 else None, k, excluded=returned)
            _lib.call("ssw_tune_topk", 3)
            got = idx.topk(q if rnd % 2 == 0 else None, k, excluded=returned)
            for a, b in zip(got, want):
                assert np.array_equal(bits(a) if a.dtype == np.float32 else a, bits(b) if b.dtype == np.float32 else b), rnd
            o = oracle.topk_images_tiebreak(oracle.scores_kernel_order(X, q), None if tiles == 1 else row2image,
                                            n_images, returned, k)
            assert np.array_equal(got[0], o[0]) and np.array_equal(got[2], o[2])
            returned.extend(got[0][:40].tolist())
            returned.extend(got[0][:3].tolist())  # repeats
        got = idx.topk(None, 5, excluded=range(n_images))
        assert got[0].shape[0] == 0
    finally:
        _lib.call("ssw_tune_topk", 3)
        idx.close()
        ref_idx.close()


def test_topk_fewer_than_k_and_all_excluded(DeviceIndex, oracle):
    X = oracle.synth_rows(2, 0, 300, 512)
    q = oracle.synth_query(0)
    idx = DeviceIndex.from_numpy(X)
    imgs, scores, rows = idx.topk(q, 1000)
    assert imgs.shape[0] == 300
    ref = oracle.topk_images_tiebreak(oracle.scores_kernel_order(X, q), None, 300, [], 1000)
    assert np.array_equal(imgs, ref[0])
    imgs, _, _ = idx.topk(q, 10, excluded=range(300))
    assert imgs.shape[0] == 0
    imgs, _, _ = idx.topk(q, 10, excluded=range(295))
    assert sorted(imgs.tolist()) == [295, 296, 297, 298, 299]
    idx.close()


def test_topk_massive_ties_take_deep_path(DeviceIndex, oracle):
    # 30000 identical rows + a few better ones: more candidates share one score than the
    # final sort can hold, so the selection must fall back to the deep radix path and
    # still return the lowest image positions among the ties.
    base = oracle.synth_rows(1, 0, 1, 512)[0]
    q = base.copy()
    X = np.repeat(base[None, :] * np.float32(0.5), 30000, axis=0)
    better = oracle.synth_rows(1, 0, 20, 512)
    better[:] = base[None, :] * np.linspace(0.6, 0.9, 20, dtype=np.float32)[:, None]
    pos = np.arange(20) * 1000 + 7
    X[pos] = better
    idx = DeviceIndex.from_numpy(X)
    for k in (10, 20, 21, 100, 4096):
        imgs, scores, rows = idx.topk(q, k)
        ref = oracle.topk_images_tiebreak(oracle.scores_kernel_order(X, q), None, 30000, [], k)
        assert np.array_equal(imgs, ref[0]), k
        assert np.array_equal(bits(scores), bits(ref[1]))
    idx.close()


@pytest.mark.parametrize("spread", [1.0, 3e-2, 1e-3, 2e-5, 3e-7])
def test_topk_of_loaded_scores_at_every_histogram_depth(DeviceIndex, oracle, spread):
    """scores spread over the whole range (one histogram level), inside one 12-bit bin (second level, taken from 1024
    first-level candidates on), inside one 24-bit prefix (more candidates than the final sort holds: deep path), with
    negative values, duplicates and -inf rows: always the exact top-k in (score desc, position asc) order"""
    rng = np.random.default_rng(int(1 / spread))
    n = 60000
    idx = DeviceIndex.synthetic(n, 512, seed=5)
    for centre in (0.37, -0.37):
        s = (centre + spread * rng.random(n)).astype(np.float32)
        s[rng.integers(0, n, 500)] = s[rng.integers(0, n, 500)]      # duplicates
        s[rng.integers(0, n, 50)] = -np.inf                          # rows that never take part
        idx.load_scores(s)
        excluded = rng.choice(n, 300, replace=False)
        for k in (1, 50, 1000, 4096):
            imgs, scores, rows = idx.topk(None, k, excluded=excluded)
            sx = s.copy()
            sx[excluded] = -np.inf
            order = np.lexsort((np.arange(n), -sx.astype(np.float64)))[:k]
            assert np.array_equal(imgs, order), (spread, centre, k)
            assert np.array_equal(bits(scores), bits(sx[order]))
            assert np.array_equal(rows, order)
    idx.close()


def test_nan_query_rejected(DeviceIndex, oracle):
    from seesaw_amd._lib import SeesawHipError
    idx = DeviceIndex.synthetic(128, 512, seed=0)
    q = oracle.synth_query(0)
    q[5] = np.nan
    with pytest.raises(SeesawHipError):
        idx.scores(q)
    idx.close()


def test_scan_of_device_generated_shard(DeviceIndex, oracle):
    # a shard generated on the device is scanned without ever touching the host
    n, first = 200000, 12345678
    idx = DeviceIndex.synthetic(n, 512, seed=42, first_row=first)
    q = oracle.synth_query(9)
    imgs, scores, rows = idx.topk(q, 100)
    X = oracle.synth_rows(42, first, n, 512)
    ref = oracle.topk_images_tiebreak(oracle.scores_kernel_order(X, q), None, n, [], 100)
    assert np.array_equal(imgs, ref[0]) and np.array_equal(bits(scores), bits(ref[1]))
    idx.close()
