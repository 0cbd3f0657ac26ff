"""enqueue / wait split of the fused round (ssw_labelprop_last_run_info [6], [7]) over the steady rounds of a knn_prop2
session at 1.56 M vectors: an A/B harness for launch-shape experiments (round 6: grids of 64 / 128, 256 / 512 and
1024 / 2048 workgroups for the frontier / row kernels wait 76, 66 and 63-64 us: the defaults stay).   python tools/perf_round_wait.py"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
from seesaw_amd.bitmap import BitMap
from seesaw_amd.label_propagation import LabelPropagation
from seesaw_amd.seesaw_bench import benchmark_loop
from seesaw_amd.seesaw_session import make_session
from seesaw_amd.synthetic import GlobalDataManager, make_dataset

matrix = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
opts = dict(matrix_options=matrix, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4, prior_weight=1.0)
ds = make_dataset("lvis", n_images=120000, tiles_per_image=13, n_categories=2, positive_frac=0.05, seed=11, knn_k=10)
ds.embedding.noise = 1.2
gdm = GlobalDataManager().add(ds)
boxes, _ = ds.load_ground_truth()
p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale"), interactive="knn_prop2", interactive_options=opts,
                  batch_size=1, shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                  start_policy="after_first_batch", index_options={"use_vec_index": False})
b = BenchParams(name="knn_prop2", ground_truth_category="c1", qstr="a c1", n_batches=30, max_results=10 ** 6)
LabelPropagation.collect_run_info = True
rec = []
_round = LabelPropagation.round


def round_rec(self, *a, **k):
    out = _round(self, *a, **k)
    rec.append((self.last_mode, self.last_frontier_us, self.last_device_wait_us, self.last_rows_recomputed))
    return out


LabelPropagation.round = round_rec
lat_all = []
for rep in range(4):
    rec.clear()
    ret = make_session(gdm, p, b=b)
    with contextlib.redirect_stdout(io.StringIO()):
        g = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
    if rep:
        lat_all.append(np.median(g["latencies"]))
inc = np.array([(e, w, r) for m, e, w, r in rec if m == 1])
print(f"incremental rounds {inc.shape[0]}, enqueue {inc[:, 0].mean():.1f} us, wait {inc[:, 1].mean():.1f} us, "
      f"enqueue + wait median {np.median(inc[:, 0] + inc[:, 1]):.1f} us, rows {inc[:, 2].mean():.0f}; median round {1e3 * np.median(lat_all):.3f} ms")
