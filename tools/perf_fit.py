"""L-BFGS fit latency of the feedback engine (MultiReg CE objective) on labelled sets of growing size."""
import time

import numpy as np

from seesaw_amd import _lib
from seesaw_amd.feedback import FbObjective, FeedbackEngine

rng = np.random.default_rng(0)
for n in (52, 208, 390):
    X = rng.standard_normal((n, 512)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    q = X[:5].mean(0)
    y = (X @ q > np.quantile(X @ q, 0.8)).astype(np.float64)
    eng = FeedbackEngine(512)
    eng.set_query(q / np.linalg.norm(q))
    eng.set_data(X, center=True)
    eng.set_targets(y, np.ones(n))
    obj = FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=_lib.SSW_FB_LOSS_CE, fit_intercept=0, reg_kind=0,
                      pos_weight=-1.0, reg_weight=0.0, margin=0.0, reg_norm_lambda=100.0, reg_data_lambda=0.0,
                      reg_query_lambda=0.0)
    w0 = (q / np.linalg.norm(q)).astype(np.float32)
    eng.fit(obj, w0, 200)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        w, info = eng.fit(obj, w0, 200)
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n}: {dt*1e3:.3f} ms per fit, {info['func_evals']} evaluations, {info['n_iter']} iterations, "
          f"{dt*1e6/info['func_evals']:.1f} us per evaluation, loss {info['loss']:.6f}", flush=True)
