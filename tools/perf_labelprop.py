"""Label-propagation sweep time on a random symmetric k-NN-like graph (GPU box)."""
import sys
import time

import numpy as np
import scipy.sparse as sp

from seesaw_amd.label_propagation import LabelPropagation

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_560_000
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 200  # fewer under a counter pass
k = 7
rng = np.random.default_rng(0)
src = np.repeat(np.arange(n, dtype=np.int64), k)
dst = (src + rng.integers(1, n, size=src.shape[0])) % n
w = rng.random(src.shape[0])
A = sp.coo_array((w, (src, dst)), shape=(n, n)).tocsr()
W = (A + A.T).tocsr()
W.sort_indices()
lp = LabelPropagation(W, reg_lambda=1.0, max_iter=1, epsilon=-1.0)
prior = np.full(n, 0.5)
ids, vals = np.arange(0, 1000, dtype=np.int64), (np.arange(1000) % 2).astype(np.float64)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):  # warm-up: lazy initialisation of the runtime, first touch of the buffers
        lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
    t1 = time.perf_counter(); lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior); t1 = time.perf_counter() - t1
    lp.max_iter = sweeps + 1
    t2 = time.perf_counter(); lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior); t2 = time.perf_counter() - t2
sweep = (t2 - t1) / sweeps
nbytes = 12.0 * W.nnz + 40.0 * n
print(f"n={n} nnz={W.nnz}: {sweep*1e3:.3f} ms per sweep, {nbytes/sweep/1e9:.0f} GB/s algorithmic, "
      f"fixed cost of a call {t1*1e3:.2f} ms", flush=True)
