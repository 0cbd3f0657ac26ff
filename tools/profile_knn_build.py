"""cProfile of the k-NN graph build as the synthetic datasets run it (development aid; GPU box)."""
import cProfile
import pstats
import sys
import time

from seesaw_amd.synthetic import make_dataset

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
ds = make_dataset("lvis", n_images=n_images, tiles_per_image=13, n_categories=2, positive_frac=0.05, seed=11, knn_k=10)
idx = None
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
g = ds.knn_graph()
pr.disable()
print("knn_graph()", time.perf_counter() - t0, "s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
