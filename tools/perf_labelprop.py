"""Label-propagation sweep time (GPU box).

  python tools/perf_labelprop.py [nodes] [sweeps] [graph] [order]
    graph: random   -- a symmetric k-NN-like graph over nothing (edges to uniformly random nodes): no locality to find
           mog      -- the exact k-NN graph (k = 10, ssw_knn_build) of mixture-of-Gaussians vectors (2000 clusters, nodes in
                       random order), symmetrised as get_weight_matrix does: the clustered case of VERDICT r2 #7
    order: none     -- nodes as given (graphs beyond 5 MiB of iterate take the column-blocked sweep)
           rcm      -- label_propagation.locality_order (reverse Cuthill-McKee), the plain sweep over the re-ordered graph
           auto     -- what the loops do: the order if locality_order() finds one worth having
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seesaw_amd.label_propagation import LabelPropagation, locality_order  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_560_000
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 200  # fewer under a counter pass
graph = sys.argv[3] if len(sys.argv) > 3 else "random"
order = sys.argv[4] if len(sys.argv) > 4 else "none"
rng = np.random.default_rng(0)
t0 = time.perf_counter()
if graph == "random":
    k = 7
    src = np.repeat(np.arange(n, dtype=np.int64), k)
    dst = (src + rng.integers(1, n, size=src.shape[0])) % n
    w = rng.random(src.shape[0])
    A = sp.coo_array((w, (src, dst)), shape=(n, n)).tocsr()
    W = (A + A.T).tocsr()
    W.sort_indices()
else:
    from seesaw_amd.knn_graph import compute_exact_knn, get_weight_matrix, rbf_kernel
    dim, n_clusters = 512, 2000
    centres = rng.standard_normal((n_clusters, dim)).astype(np.float32)
    lab = rng.integers(0, n_clusters, n)  # nodes in random order: cluster membership says nothing about the id
    X = np.empty((n, dim), dtype=np.float32)
    for a in range(0, n, 1 << 17):  # chunks: 1.56 M x 512 normals at once would double the footprint
        b = min(n, a + (1 << 17))
        X[a:b] = centres[lab[a:b]] + 0.3 * rng.standard_normal((b - a, dim), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    df = compute_exact_knn(X, 10)
    del X
    W = get_weight_matrix(df, kfun=rbf_kernel(0.05), self_edges=False, normalized=False, symmetric=True, device=0)
    W = sp.csr_matrix(W)
    W.sort_indices()
    rows = np.repeat(np.arange(n), np.diff(W.indptr))
    print(f"k-NN graph of {n} mixture-of-Gaussians vectors: nnz {W.nnz}, edges inside a cluster "
          f"{(lab[rows] == lab[W.indices]).mean():.3f}", flush=True)
t_graph = time.perf_counter() - t0
node_order, t_order = None, 0.0
if order in ("rcm", "auto"):
    t0 = time.perf_counter()
    node_order = locality_order(W) if order == "auto" else locality_order(W, min_gain=0.0, min_nodes=0)
    if order == "rcm" and node_order is None:  # declined (unclustered graph): force it, to show what it would cost
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        pat = sp.csr_matrix((np.ones(W.nnz, dtype=np.int8), W.indices, W.indptr), shape=W.shape)
        node_order = np.empty(n, dtype=np.int32)
        node_order[np.asarray(reverse_cuthill_mckee((pat + pat.T).tocsr(), symmetric_mode=True))] = np.arange(n, dtype=np.int32)
    t_order = time.perf_counter() - t0
lp = LabelPropagation(W, reg_lambda=1.0, max_iter=1, epsilon=-1.0, node_order=node_order)
prior = np.full(n, 0.5)
ids, vals = np.arange(0, 1000, dtype=np.int64), (np.arange(1000) % 2).astype(np.float64)
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):  # warm-up: lazy initialisation of the runtime, first touch of the buffers
        lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
    t1 = time.perf_counter(); lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior); t1 = time.perf_counter() - t1
    lp.max_iter = sweeps + 1
    t2 = time.perf_counter(); out = lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior); t2 = time.perf_counter() - t2
sweep = (t2 - t1) / sweeps
nbytes = 12.0 * W.nnz + 40.0 * n
print(f"graph {graph}, order {order if node_order is not None else 'none' + (' (declined)' if order == 'auto' else '')}: "
      f"n={n} nnz={W.nnz}: {sweep*1e3:.3f} ms per sweep, {nbytes/sweep/1e9:.0f} GB/s algorithmic, "
      f"fixed cost of a call {t1*1e3:.2f} ms; graph {t_graph:.1f} s, ordering {t_order:.2f} s; "
      f"checksum {float(np.sum(out)):.12f}", flush=True)
