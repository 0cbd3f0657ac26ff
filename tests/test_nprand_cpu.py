"""ssw_np_permutation_prefix (host-only C, csrc/nprand.hip) against numpy itself: the same first k entries of
np.random.permutation(n) and the same stream afterwards, for sizes on both sides of every power of two the rejection
sampler's masks step at, with the generator at arbitrary positions inside its 624-word block."""
import numpy as np
import pytest

from seesaw_amd.nprand import permutation_prefix


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 31, 32, 33, 624, 625, 1000, 65535, 65536, 65537, 300001])
def test_prefix_and_stream_equal_numpy(n):
    for seed, burn in ((0, 0), (1, 7), (12345, 623), (99, 624), (7, 1250)):
        np.random.seed(seed)
        np.random.randint(0, 10, size=burn)  # move the position inside / across the 624-word block
        state = np.random.get_state()
        ref = np.random.permutation(n)
        ref_next = np.random.random(5)
        for k in (0, 1, 10, n, n + 3):
            np.random.set_state(state)
            got = permutation_prefix(n, k)
            assert got.dtype == np.int64
            assert np.array_equal(got, ref[:k]), (n, seed, burn, k)
            assert np.array_equal(np.random.random(5), ref_next), "the stream must continue where numpy's call leaves it"


@pytest.mark.parametrize("n,k", [(300001, 1000), (300001, 18750), (300001, 18751), (160000, 10000), (1000, 62), (1000, 63),
                                 (64, 4), (17, 1), (16, 1)])
def test_prefixes_on_both_sides_of_the_trace_threshold(n, k):
    """prefixes up to n/16 are followed backwards through the swaps (occupancy bitmap + pointer-at-position array, kept
    per thread between calls), longer ones shuffle the array: both against numpy, twice in a row on one stream"""
    np.random.seed(n + k)
    ref = [np.random.permutation(n)[:k] for _ in range(2)]
    ref_next = np.random.random(3)
    np.random.seed(n + k)
    got = [permutation_prefix(n, k) for _ in range(2)]
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(np.random.random(3), ref_next)


def test_makexy_rows_draw_is_the_reference_expression():
    from seesaw_amd.loops.util import makeXy_rows

    class _Ranker:
        def __init__(self, n):
            self.is_labeled = np.zeros(n)
            self.is_labeled[[3, 17, 40]] = 1
            self.labels = np.zeros(n)
            self.labels[17] = 1
            self._s = np.linspace(0, 1, n)

        def current_scores(self):
            return self._s

    lr = _Ranker(5000)
    np.random.seed(4)
    rows, y, is_real = makeXy_rows(lr, sample_size=100)
    np.random.seed(4)
    unl = np.nonzero(~(lr.is_labeled > 0))[0]
    pick = unl[np.random.permutation(unl.shape[0])[:100]]
    assert np.array_equal(rows, np.concatenate(([3, 17, 40], pick)))
    assert np.array_equal(y, np.concatenate(([0, 1, 0], lr._s[pick])))
    assert is_real.sum() == 3


def test_makexy_rows_from_the_label_map_is_the_array_path():
    """rankers that keep `_label_map` get the unlabelled rows by arithmetic on the sorted labelled rows"""
    from seesaw_amd.loops.util import makeXy_rows

    class _Ranker:
        def __init__(self, n, labelled, with_map):
            self.is_labeled = np.zeros(n)
            self.labels = np.zeros(n)
            rng = np.random.default_rng(5)
            for i in labelled:
                self.is_labeled[i] = 1
                self.labels[i] = float(rng.integers(0, 2))
            if with_map:
                self._label_map = {int(i): float(self.labels[i]) for i in labelled}
            self._s = np.linspace(0, 1, n)

        def current_scores(self):
            return self._s

    for labelled in ([0], [4999], [0, 1, 2, 3], [7, 8, 9, 4998, 4999], list(range(0, 5000, 7)), [2500], []):
        for k in (1, 100, 4900, 6000):
            out = []
            for with_map in (False, True):
                np.random.seed(9)
                out.append(makeXy_rows(_Ranker(5000, labelled, with_map), sample_size=k))
            for a, b in zip(*out):
                assert np.array_equal(a, b), (labelled, k)


def test_generator_gaussian_cache_survives():
    """has_gauss / cached_gaussian of the legacy state tuple are passed through untouched"""
    np.random.seed(3)
    np.random.standard_normal(1)  # leaves one cached gaussian
    st = np.random.get_state()
    assert st[3] == 1
    ref = np.random.permutation(50)
    ref_n = np.random.standard_normal(2)
    np.random.set_state(st)
    assert np.array_equal(permutation_prefix(50, 50), ref)
    assert np.array_equal(np.random.standard_normal(2), ref_n)


def test_scalar_draw_path_in_a_process_without_avx512():
    """the word-wise draws have an AVX-512 variant chosen at run time; SSW_NPRAND_NO_AVX512 forces the scalar one"""
    import os
    import subprocess
    import sys
    code = ("import numpy as np; from seesaw_amd.nprand import permutation_prefix\n"
            "for n, k in ((300001, 1000), (70000, 70000), (1560, 97), (65537, 10)):\n"
            "    np.random.seed(n); ref = np.random.permutation(n)[:k]; nxt = np.random.random(2)\n"
            "    np.random.seed(n); got = permutation_prefix(n, k)\n"
            "    assert np.array_equal(got, ref) and np.array_equal(np.random.random(2), nxt), (n, k)\n"
            "print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSW_NPRAND_NO_AVX512="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
