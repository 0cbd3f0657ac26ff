"""Where the PYTHON share of a knn_prop2 round goes: mean us per call of the methods on the refine() / next() path over the
steady rounds of a session at 1.56 M vectors (wrappers add ~0.3 us each).   python tools/round_python.py [images]"""
import contextlib
import io
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from seesaw_amd import _lib, seesaw_bench, seesaw_session
from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
from seesaw_amd.bitmap import BitMap
from seesaw_amd.indices.multiscale import multiscale_index as mi
from seesaw_amd.label_propagation import LabelPropagation
from seesaw_amd.loops import graph_based, loop_base
from seesaw_amd.research import knn_methods
from seesaw_amd.seesaw_bench import benchmark_loop
from seesaw_amd.seesaw_session import make_session
from seesaw_amd.synthetic import GlobalDataManager, make_dataset

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
matrix = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
opts = dict(matrix_options=matrix, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4, prior_weight=1.0)
ds = make_dataset("lvis", n_images=n_images, tiles_per_image=13, n_categories=2, positive_frac=0.05, seed=11, knn_k=10)
ds.embedding.noise = 1.2
gdm = GlobalDataManager().add(ds)
boxes, _ = ds.load_ground_truth()
p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale"), interactive="knn_prop2", interactive_options=opts,
                  batch_size=1, shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                  start_policy="after_first_batch", index_options={"use_vec_index": False})
b = BenchParams(name="knn_prop2", ground_truth_category="c1", qstr="a c1", n_batches=30, max_results=10 ** 6)
acc = defaultdict(lambda: [0.0, 0])


def wrap(owner, name, label=None):
    orig = getattr(owner, name)
    label = label or f"{getattr(owner, '__name__', type(owner).__name__)}.{name}"

    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            e = acc[label]
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(owner, name, f)


for rep in range(3):
    ret = make_session(gdm, p, b=b)
    if rep == 2:
        for owner, name in ((seesaw_session.Session, "next"), (seesaw_session.Session, "refine"), (seesaw_session.Session, "last_batch"),
                            (seesaw_session.Session, "update_last_batch"), (seesaw_bench, "fill_imdata"),
                            (graph_based.KnnProp2, "refine"), (graph_based.KnnProp2, "next_batch"),
                            (loop_base.LoopBase, "refine_external"), (loop_base.LoopBase, "next_batch_external"),
                            (mi.BoxFeedbackQuery, "getXy"), (mi.BoxFeedbackQuery, "_matched_arrays"),
                            (mi.MultiscaleIndex, "topk_after_update"), (mi.MultiscaleIndex, "_excluded_positions"),
                            (mi.MultiscaleIndex, "_activations_from_best"),
                            (knn_methods.LabelPropagationRanker2, "update_and_select"),
                            (knn_methods.BaseLabelPropagationRanker, "update_labels"),
                            (knn_methods.BaseLabelPropagationRanker, "_sorted_label_ids"),
                            (knn_methods.BaseLabelPropagationRanker, "_refresh_has_negative"),
                            (LabelPropagation, "round"), (_lib, "call")):
            wrap(owner, name)
    with contextlib.redirect_stdout(io.StringIO()):
        g = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
lat = np.asarray(g["latencies"])
print(f"knn_prop2 {n_images} images: mean {1e3 * lat.mean():.3f} ms, median {1e3 * np.median(lat):.3f} ms (with the wrappers)")
for label, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {label:55s} {n:4d} calls  {1e6 * t / n:8.1f} us per call  {1e6 * t / 29:8.1f} us per round")
