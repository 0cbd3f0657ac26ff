"""Host side of the locality-ordered label-propagation layout (seesaw_amd/label_propagation.py: locality_order,
ordered_csr): pure numpy / scipy, no GPU.  The device side is tests/test_labelprop_gpu.py."""
import numpy as np
import scipy.sparse as sp

from seesaw_amd.label_propagation import locality_order, ordered_csr


def test_ordered_csr_keeps_each_rows_entries_in_original_column_order():
    rng = np.random.default_rng(0)
    n = 300
    W = sp.random(n, n, density=0.05, random_state=1, format="csr")
    W.sort_indices()
    wsum = np.asarray(W.sum(0)).reshape(-1)
    perm = rng.permutation(n).astype(np.int32)
    indptr, indices, data, ws = ordered_csr(W, wsum, perm)
    old_of_new = np.argsort(perm)
    assert indptr[0] == 0 and indptr[-1] == W.nnz
    for r in range(n):
        i = old_of_new[r]
        a, b = W.indptr[i], W.indptr[i + 1]
        assert np.array_equal(indices[indptr[r]:indptr[r + 1]], perm[W.indices[a:b]])  # relabelled, original order
        assert np.array_equal(data[indptr[r]:indptr[r + 1]], W.data[a:b])
    assert np.array_equal(ws, wsum[old_of_new])


def test_locality_order_is_taken_on_clustered_graphs_and_declined_elsewhere():
    rng = np.random.default_rng(0)
    n, k, nc = 40000, 8, 200
    lab = rng.integers(0, nc, n)
    members = [np.nonzero(lab == c)[0] for c in range(nc)]
    rows = np.repeat(np.arange(n), k)
    clustered = np.concatenate([rng.choice(members[lab[i]], k) for i in range(n)])
    Wc = sp.csr_matrix((np.ones(n * k), (rows, clustered)), shape=(n, n))
    Wc = (Wc + Wc.T).tocsr()
    order = locality_order(Wc, min_nodes=1000, window=2048)
    assert order is not None and order.dtype == np.int32 and np.array_equal(np.sort(order), np.arange(n))
    r = np.repeat(np.arange(n), np.diff(Wc.indptr))
    assert (np.abs(order[r].astype(np.int64) - order[Wc.indices]) < 2048).mean() > 0.9
    Wr = sp.csr_matrix((np.ones(n * k), (rows, rng.integers(0, n, n * k))), shape=(n, n))
    Wr = (Wr + Wr.T).tocsr()
    assert locality_order(Wr, min_nodes=1000, window=2048) is None   # nothing to find
    assert locality_order(Wc) is None                                # under 2^19 nodes the iterate stays in L2
