"""cProfile of one benchmark_loop session (development aid)."""
import cProfile
import contextlib
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
from seesaw_amd.bitmap import BitMap
from seesaw_amd.seesaw_bench import benchmark_loop
from seesaw_amd.seesaw_session import make_session
from seesaw_amd.synthetic import GlobalDataManager, make_dataset

name = sys.argv[1] if len(sys.argv) > 1 else "multi_reg"
n_images = int(sys.argv[2]) if len(sys.argv) > 2 else 1109
agg = sys.argv[3] if len(sys.argv) > 3 else "plain_score"  # or avg_score (aug_larger all: the reference's standard config)
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
for rep in range(2):
    ret = make_session(gdm, p, b=b)
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable()
        g = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
        pr.disable()
print("mean latency ms", 1e3 * np.mean(g["latencies"]))
print("per-round ms", [round(1e3 * v, 3) for v in g["latencies"]])
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SSW_PROFILE_SORT", "cumulative")).print_stats(int(os.environ.get("SSW_PROFILE_ROWS", "35")))
