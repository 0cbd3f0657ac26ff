"""ssw_np_permutation_prefix_dev: the draw whose walk through the swaps runs on the GPU (csrc/nprand.hip, k_np_trace)
against numpy itself -- same prefix, same stream afterwards -- on both sides of its size thresholds."""
import numpy as np
import pytest

from seesaw_amd.nprand import permutation_prefix

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k", [(1 << 18, 1), (1 << 18, 1000), ((1 << 18) + 1, 16384), (300001, 18750), (1559883, 10000),
                                 (1559883, 3), (2000003, 60000), ((1 << 18) - 1, 100), (300001, 18751), (1000, 10)])
def test_device_walk_equals_numpy(n, k):
    for seed, burn in ((n + k, 0), (5, 623), (6, 1250)):
        np.random.seed(seed)
        np.random.randint(0, 10, size=burn)
        state = np.random.get_state()
        ref = [np.random.permutation(n)[:k] for _ in range(2)]  # twice in a row on one stream: the scratch is reused
        ref_next = np.random.random(4)
        np.random.set_state(state)
        got = [permutation_prefix(n, k, device=0) for _ in range(2)]
        assert got[0].dtype == np.int64
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (n, k, seed)
        assert np.array_equal(np.random.random(4), ref_next), "the stream must continue where numpy's call leaves it"


def test_device_and_host_forms_agree_on_a_shrinking_pool():
    """PseudoLR's rounds: the unlabelled pool shrinks by a few rows each round, the draws follow each other on one stream"""
    np.random.seed(77)
    state = np.random.get_state()
    sizes = [1560000 - 13 * r for r in range(6)]
    host = [permutation_prefix(n, 10000) for n in sizes]
    np.random.set_state(state)
    dev = [permutation_prefix(n, 10000, device=0) for n in sizes]
    for a, b in zip(host, dev):
        assert np.array_equal(a, b)
