"""Where a feedback round's latency goes: per round, the time inside session.next(), the benchmark's
bookkeeping and session.refine(), and inside them every C-ABI call with its duration.  Prints round 1 (the slow one),
round 2 and the median round.   python tools/round_stamps.py knn_prop2 120000 [avg_score]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seesaw_amd import _lib
from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
from seesaw_amd.bitmap import BitMap
from seesaw_amd.seesaw_bench import benchmark_loop
from seesaw_amd.seesaw_session import make_session
from seesaw_amd.synthetic import GlobalDataManager, make_dataset

name = sys.argv[1] if len(sys.argv) > 1 else "knn_prop2"
n_images = int(sys.argv[2]) if len(sys.argv) > 2 else 120000
agg = sys.argv[3] if len(sys.argv) > 3 else "plain_score"
matrix = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
opts = {"plain": None,
        "multi_reg": dict(label_loss_type="ce_loss", rank_loss_margin=0.2, use_qvec_norm=None, reg_data_lambda=0.0,
                          reg_norm_lambda=100.0, reg_query_lambda=0.0, verbose=False, max_iter=200, pos_weight="balanced",
                          lr=1.0, matrix_options=matrix),
        "knn_prop2": dict(matrix_options=matrix, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0,
                          calib_b=-0.4, prior_weight=1.0),
        "pseudo_lr": dict(switch_over=True, real_sample_weight=1.0, sample_size=10000,
                          log_reg_params=dict(class_weights=1.0, scale="centered", reg_lambda=1.0, max_iter=200.0, lr=1,
                                              fit_intercept=False),
                          label_prop_params=dict(matrix_options=matrix, normalize_scores=False,
                                                 sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4,
                                                 prior_weight=1.0))}[name]
ds = make_dataset("lvis", n_images=n_images, tiles_per_image=13, n_categories=2, positive_frac=0.05, seed=11,
                  knn_k=10 if name in ("knn_prop2", "pseudo_lr") else 0)
ds.embedding.noise = 1.2
gdm = GlobalDataManager().add(ds)
boxes, _ = ds.load_ground_truth()
p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale"), interactive=name, interactive_options=opts,
                  batch_size=1, shortlist_size=50, agg_method=agg, aug_larger="greater" if agg == "plain_score" else "all",
                  start_policy="after_first_batch", index_options={"use_vec_index": False})
b = BenchParams(name=name, ground_truth_category="c1", qstr="a c1", n_batches=30, max_results=10 ** 6)

events = []  # (t_start, dt, label)
orig_call = _lib.call


def traced_call(fn, *a):
    t0 = time.perf_counter()
    try:
        return orig_call(fn, *a)
    finally:
        events.append((t0, time.perf_counter() - t0, fn))


def wrap_method(obj, meth, label):
    orig = getattr(obj, meth)

    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            events.append((t0, time.perf_counter() - t0, label))
    setattr(obj, meth, f)


from seesaw_amd.label_propagation import LabelPropagation
_round = LabelPropagation.round


def round_with_info(self, *a, **k):
    out = _round(self, *a, **k)
    events.append((time.perf_counter(), 0.0, f"   lp info: mode {self.last_mode}, enqueue {self.last_frontier_us:.1f} us, wait {self.last_device_wait_us:.1f} us, "
                                             f"launches {self.last_launches}, rows {self.last_rows_recomputed}"))
    return out


from seesaw_amd.feedback import FeedbackEngine
_fit = FeedbackEngine.fit


def fit_with_info(self, *a, **k):
    t0 = time.perf_counter()
    w, info = _fit(self, *a, **k)
    dt = time.perf_counter() - t0
    events.append((time.perf_counter(), 0.0, f"   fit info: rows {self.n}, evals {info['func_evals']}, iterations {info['n_iter']}, "
                                             f"{1e6 * dt / max(1, info['func_evals']):.1f} us per evaluation, on device {info['on_device']}"))
    return w, info


for rep in range(3):
    ret = make_session(gdm, p, b=b)
    sess = ret["session"]
    if rep == 2:
        LabelPropagation.collect_run_info = True
        FeedbackEngine.fit = fit_with_info
        LabelPropagation.round = round_with_info
        _lib.call = traced_call
        for mod in list(sys.modules.values()):  # modules that bound `call` by name
            if getattr(mod, "__name__", "").startswith("seesaw_amd") and getattr(mod, "_lib", None) is _lib:
                pass
        wrap_method(sess, "next", "== next")
        wrap_method(sess, "refine", "== refine")
    with contextlib.redirect_stdout(io.StringIO()):
        t_loop = time.perf_counter()
        g = benchmark_loop(session=sess, box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
_lib.call = orig_call
lat = np.asarray(g["latencies"])
print(f"{name} {n_images} images: mean {1e3 * lat.mean():.3f} ms, median {1e3 * np.median(lat):.3f}, slowest {1e3 * lat.max():.3f} (round {int(lat.argmax()) + 1})")
print("per-round ms", [round(1e3 * v, 3) for v in lat])
# split the events by round: a round starts at its "== next" event
starts = [t for t, _, lab in events if lab == "== next"]
order = np.argsort(lat)
show = sorted({0, 1, int(order[len(order) // 2])})
for r in show:
    t_a = starts[r]
    t_b = starts[r + 1] if r + 1 < len(starts) else float("inf")
    evs = sorted(e for e in events if t_a <= e[0] < t_b)
    print(f"--- round {r + 1}: latency {1e3 * lat[r]:.3f} ms" if r < len(lat) else f"--- round {r + 1}")
    c_total = 0.0
    for t, dt, lab in evs:
        if not lab.startswith("=="):
            c_total += dt
        print(f"   +{1e3 * (t - t_a):7.3f} ms  {1e3 * dt:7.3f} ms  {lab}")
    print(f"   C-ABI calls {1e3 * c_total:.3f} ms in {sum(1 for e in evs if not e[2].startswith('=='))} calls")
