"""Fuzz: the one-launch fit against the host-driven fit on random problems (development aid; GPU box).
Prints the first mismatch (iteration / evaluation counts, coefficient bits) or a summary."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seesaw_amd import _lib  # noqa: E402
from seesaw_amd.feedback import FbObjective, FeedbackEngine  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
eng = FeedbackEngine(512)
bad = 0
evals = 0
for t in range(trials):
    n = int(rng.integers(0, 400)) if t % 8 else int(rng.integers(400, 1025))
    X = rng.standard_normal((n, 512)).astype(np.float32)
    if n:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        X += rng.standard_normal(512).astype(np.float32) * float(rng.uniform(0, 0.3))
    q = rng.standard_normal(512).astype(np.float32)
    y = (rng.uniform(size=n) > rng.uniform(0.2, 0.9)).astype(np.float64)
    eng.set_query(q)
    eng.set_data(X, center=bool(n))
    eng.set_targets(y, rng.uniform(0.2, 2.0, n) if n else None)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        obj = FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=0, fit_intercept=0, reg_kind=0, pos_weight=-1.0, reg_weight=0.0,
                          margin=0.2, reg_norm_lambda=float(rng.choice([1.0, 100.0, 1000.0])), reg_data_lambda=0.0,
                          reg_query_lambda=float(rng.choice([0.0, 1.0, 100.0])))
    elif kind == 1 and 2 <= n <= 300:
        obj = FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=int(rng.integers(1, 3)), fit_intercept=0, reg_kind=0,
                          pos_weight=-1.0, reg_weight=0.0, margin=0.2, reg_norm_lambda=100.0, reg_data_lambda=0.0,
                          reg_query_lambda=10.0)
    else:
        obj = FbObjective(kind=_lib.SSW_FB_LOGREG, loss_type=0, fit_intercept=int(rng.integers(0, 2)),
                          reg_kind=int(rng.integers(0, 4)), pos_weight=float(rng.uniform(0.5, 5)),
                          reg_weight=float(rng.uniform(0, 2)) / max(n, 1), margin=0, reg_norm_lambda=0,
                          reg_data_lambda=0, reg_query_lambda=0)
    P = 512 + (1 if (obj.kind == _lib.SSW_FB_LOGREG and obj.fit_intercept) else 0)
    w0 = (rng.standard_normal(P) * float(rng.uniform(0.01, 1.0))).astype(np.float32)
    mi = int(rng.choice([5, 40, 200]))
    try:
        wd, idev = eng.fit(obj, w0, mi)
        os.environ["SSW_FB_HOST_DRIVER"] = "1"
        try:
            wh, ihost = eng.fit(obj, w0, mi)
        finally:
            del os.environ["SSW_FB_HOST_DRIVER"]
    except _lib.SeesawHipError as e:
        print("trial", t, "raised", e)
        os.environ.pop("SSW_FB_HOST_DRIVER", None)
        continue
    evals += idev["func_evals"]
    same = (idev["n_iter"], idev["func_evals"]) == (ihost["n_iter"], ihost["func_evals"]) and \
        np.array_equal(wd.view(np.uint32), wh.view(np.uint32))
    if not same:
        bad += 1
        print(f"MISMATCH trial {t}: n={n} kind={obj.kind} loss_type={obj.loss_type} reg_kind={obj.reg_kind} "
              f"dev={idev} host={ihost} max|dw|={np.abs(wd - wh).max():.3e}")
print(f"{trials} trials, {evals} evaluations on the device path, {bad} mismatches")
