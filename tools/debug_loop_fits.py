"""The session's own multi_reg fits (tests/golden/bench_loop.npz: multi_reg_fit*): HIP fit vs the reference's
coefficients, in rank scores over ALL vectors of the dataset.  SSW_FB_EXACT_LOSS=1 switches to exact f64 losses."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seesaw_amd import _lib  # noqa: E402
from seesaw_amd.feedback import FeedbackEngine  # noqa: E402
from seesaw_amd.synthetic import make_dataset  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "bench_loop.npz"))
ds = make_dataset("lvis", knn_k=0, **json.loads(str(g["datasets"]))["A"]["make"])
eng = FeedbackEngine(512)
obj = _lib.FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=0, fit_intercept=0, reg_kind=0, pos_weight=-1.0,
                       reg_weight=0.0, margin=0.2, reg_norm_lambda=100.0, reg_data_lambda=0.0, reg_query_lambda=0.0)
print("mode:", "exact f64" if os.environ.get("SSW_FB_EXACT_LOSS") else "torch rounding")
for r in range(int(g["multi_reg_n_fits"])):
    rows, y, img, q = (g[f"multi_reg_fit{r}_{k}"] for k in ("rows", "y", "img", "q"))
    ref = g[f"multi_reg_fit{r}_coeff"]
    _, inv, counts = np.unique(img, return_inverse=True, return_counts=True)
    eng.set_data(ds.vectors[rows], center=True)
    eng.set_targets(y, 1.0 / counts[inv])
    eng.set_query(q)
    w, info = eng.fit(obj, q / np.linalg.norm(q), max_iter=200)
    coeff = w[:512] / np.linalg.norm(w[:512])
    d = np.abs(ds.vectors @ (coeff.astype(np.float64) - ref)).max()
    n_ref = g[f"multi_reg_fit{r}_traj_w"].shape[0] if f"multi_reg_fit{r}_traj_w" in g.files else -1
    print(f"fit {r}: n={rows.shape[0]} evals={info['func_evals']} (reference {n_ref}) loss={info['loss']:.8f} "
          f"|scores - reference| over all vectors = {d:.2e}")
