"""Where does the HIP L-BFGS path leave the reference's?  For a golden case, fit with max_iter = 1, 2, ... and
match every iterate against the closure trajectory the reference recorded.
  python tools/debug_fit_path.py logreg 3 | multireg 4 | loop 3   (loop = the session fits in bench_loop.npz)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seesaw_amd import _lib  # noqa: E402
from seesaw_amd.feedback import FeedbackEngine  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def walk(eng, obj, w0, W, L, ref, Xall=None):
    print("reference closure losses:", [f"{v:.9f}" for v in L[:40]])
    last = None
    for k in range(1, 120):
        w, info = eng.fit(obj, w0, max_iter=k)
        d = np.abs(W - w[None, :512]).max(axis=1)
        t = int(np.argmin(d))
        wn = w[:512] / np.linalg.norm(w[:512]) if ref is not None and abs(np.linalg.norm(ref) - 1) < 1e-4 else w[:512]
        line = (f"max_iter={k}: n_iter={info['n_iter']} evals={info['func_evals']} loss={info['loss']:.9f} nearest "
                f"reference eval #{t} (|dw| = {d[t]:.2e}); |coeff - ref| = {np.abs(wn - ref.reshape(-1)).max():.2e}")
        print(line)
        if info["n_iter"] < k - 3:
            break


def main(kind, case):
    if kind == "logreg":
        g = np.load(os.path.join(GOLDEN, "logreg.npz"))
        c = case
        X, y, q = g[f"c{c}_X"], g[f"c{c}_y"], g[f"c{c}_q"]
        cw, sw, n = float(g[f"c{c}_cw"]), g[f"c{c}_sw"], X.shape[0]
        pw = max(int((y == 0).sum()), 1) / max(int((y == 1).sum()), 1) if cw < 0 else cw
        eng = FeedbackEngine(512)
        eng.set_data(X, center=True)
        eng.set_targets(y, None if sw.size == 0 else sw)
        eng.set_query(q)
        obj = _lib.FbObjective(kind=_lib.SSW_FB_LOGREG, loss_type=0, fit_intercept=0, reg_kind=_lib.SSW_FB_REG_VECTOR,
                               pos_weight=pw, reg_weight=float(g[f"c{c}_lam"]) / n, margin=0, reg_norm_lambda=0,
                               reg_data_lambda=0, reg_query_lambda=0)
        walk(eng, obj, g[f"c{c}_w0"].reshape(-1), g[f"c{c}_traj_w"], g[f"c{c}_traj_loss"], g[f"c{c}_coeff"])
        return
    code = {"ce_loss": 0, "pairwise_rank_loss": 1, "pairwise_logistic_loss": 2}
    if kind == "multireg":
        g = np.load(os.path.join(GOLDEN, "multireg.npz"))
        c = case
        X, y, img, q = g[f"c{c}_X"], g[f"c{c}_y"], g[f"c{c}_img"], g[f"c{c}_q"]
        lt, dl, ql, xlx = str(g[f"c{c}_loss_type"]), float(g[f"c{c}_data_lam"]), float(g[f"c{c}_query_lam"]), g["xlx"]
        W, L, ref = g[f"c{c}_traj_w"], g[f"c{c}_traj_loss"], g[f"c{c}_coeff"]
    else:
        from seesaw_amd.synthetic import make_dataset
        g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
        spec = json.loads(str(g["datasets"]))["A"]
        ds = make_dataset("lvis", knn_k=0, **spec["make"])
        r = case
        rows = g[f"multi_reg_fit{r}_rows"]
        X, y, img, q = ds.vectors[rows], g[f"multi_reg_fit{r}_y"], g[f"multi_reg_fit{r}_img"], g[f"multi_reg_fit{r}_q"]
        lt, dl, ql, xlx = "ce_loss", 0.0, 0.0, None
        W, L, ref = g[f"multi_reg_fit{r}_traj_w"], g[f"multi_reg_fit{r}_traj_loss"], g[f"multi_reg_fit{r}_coeff"]
    eng = FeedbackEngine(512)
    if xlx is not None:
        eng.set_xlx(xlx)
    _, inv, counts = np.unique(img, return_inverse=True, return_counts=True)
    eng.set_data(X, center=True)
    eng.set_targets(y, 1.0 / counts[inv])
    eng.set_query(q)
    obj = _lib.FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=code[lt], fit_intercept=0, reg_kind=0, pos_weight=-1.0,
                           reg_weight=0.0, margin=0.2, reg_norm_lambda=100.0, reg_data_lambda=dl, reg_query_lambda=ql)
    walk(eng, obj, q / np.linalg.norm(q), W, L, ref)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
