import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_lknn_cpu import session_graph
from seesaw_amd.loops.LKNN_model import LKNNModel, initial_gamma_array
from seesaw_amd.research.active_search.common import Dataset
g = np.load(os.path.join(ROOT, "tests", "golden", "lknn.npz"))
N, D, nbr, W, truth = session_graph(g)
model = LKNNModel.from_dataset(Dataset.from_vectors(np.zeros((N, 1))), weight_matrix=W, gamma=initial_gamma_array(0.1, N))
for h in (2, 9):
    _, _, vals = model.top_sum(K=h - 1, return_values=True)
    ref = g[f"h{h}_values_r0"]
    bad = np.nonzero(vals.view(np.uint64) != ref.view(np.uint64))[0]
    print(h, "mismatches", bad.shape[0], "of", N, "max abs diff", np.abs(vals - ref).max(), "ulps", (vals.view(np.int64) - ref.view(np.int64))[bad][:10], bad[:10])
    i = int(bad[0]) if bad.size else 0
    numer = model.numerators + model.gamma; denom = model.denominators + 1
    s = numer / denom
    print(" node", i, "s", repr(s[i]), "nbrs", np.sort(nbr[i]), "ours", repr(vals[i]), "ref", repr(ref[i]))
