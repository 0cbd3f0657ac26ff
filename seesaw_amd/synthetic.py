"""In-memory synthetic datasets so Session / benchmark_loop run without a seesaw root on disk.

The reference reads its datasets (file_meta.parquet, ground_truth/box_data.parquet, index
parquet files, knn graphs) from a shared filesystem through Ray (seesaw/dataset.py,
dataset_manager.py -- out of scope).  BASELINE.json's configs are
synthetic, so this module supplies objects with the same methods the session / bench layers
call (`get_dataset`, `load_index`, `load_ground_truth`, `load_subset`, `get_urls`,
`file_meta`, `load_eval_categories`) backed by generated vectors:

* every category c has a hidden unit direction h_c; a positive image has one tile pulled
  towards h_c, and its ground-truth box is that tile's box;
* the "text embedding" of the query `"a <category>"` is h_c blurred with noise (standing in
  for CLIP's text tower, which seesaw_amd.models provides with random-init weights);
* BASELINE config C1 (10k x 512, one vector per image, top-100, `plain`) is `make_c1()`;
  the LVIS-shape loop (13 tiles per image, batch 1, shortlist 50) is `make_lvis_shape()`.
"""
from __future__ import annotations

import zlib

import numpy as np
import pandas as pd

from .knn_graph import MAX_EXACT_K, KNNGraph, compute_exact_knn


def _unit(x):
    return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)


class SyntheticEmbedding:
    """from_string("a <category>") -> [1, dim] vector near the category's hidden direction."""

    def __init__(self, directions: dict, dim: int, noise: float = 0.35, seed: int = 0):
        self.directions = directions
        self.dim = dim
        self.noise = noise
        self.seed = seed
        self.string_cache = {}

    def from_string(self, *, string=None, str_vec=None, numpy=True):
        if str_vec is not None:
            return str_vec
        if string in self.string_cache:
            return self.string_cache[string]
        key = string[2:] if string.startswith("a ") else string
        rng = np.random.default_rng(zlib.crc32(f"{self.seed}:{string}".encode()))  # stable across processes
        base = self.directions.get(key)
        noise = _unit(rng.standard_normal(self.dim))
        vec = noise if base is None else base + self.noise * noise
        out = _unit(vec).reshape(1, -1)
        self.string_cache[string] = out
        return out


class SyntheticDataset:
    def __init__(self, name: str, *, vectors: np.ndarray, vector_meta: pd.DataFrame, box_data: pd.DataFrame,
                 categories, directions: dict, image_size=(640, 480), knn_k: int = 0, device: int = 0,
                 embedding=None):
        self.name = name
        self.path = f"synthetic://{name}"
        self.vectors = vectors
        self.vector_meta = vector_meta
        self.box_data = box_data
        self.categories = list(categories)
        self.directions = directions
        self.device = device
        self.image_size = image_size
        dbidx = np.unique(vector_meta.dbidx.values)
        self.file_meta = pd.DataFrame({"file_path": [f"{name}/{d:08d}.jpg" for d in dbidx]},
                                      index=pd.Index(dbidx, name="dbidx"))
        self.paths = self.file_meta.file_path.values
        self.embedding = embedding or SyntheticEmbedding(directions, vectors.shape[1])
        self.knn_k = knn_k
        self._index = None
        self._knng = None

    # ---- what Session / benchmark_loop call ---------------------------------------------
    def load_subset(self, c_name):
        return self

    def load_eval_categories(self):
        return list(self.categories)

    def load_ground_truth(self):
        """(box_data, qgt): qgt[c] = 1 for images with a box of category c, else 0."""
        ids = self.file_meta.index.values
        qgt = pd.DataFrame(0.0, index=pd.Index(ids, name="dbidx"), columns=self.categories)
        for c in self.categories:
            qgt.loc[self.box_data[self.box_data.category == c].dbidx.unique(), c] = 1.0
        return self.box_data, qgt

    def get_urls(self, idxbatch):
        return [f"/data/{self.name}/{int(i):08d}.jpg" for i in idxbatch]

    def knn_graph(self, name="exact") -> KNNGraph:
        if self._knng is None:
            assert self.knn_k > 0, "dataset built without a k-NN graph"
            dev = getattr(self._index, "_dev", None) if self._index is not None else None  # matrix already resident
            # stored like the reference's graphs: a pool larger than the k the loops keep, so that
            # KNNGraph.restrict_k(k=knn_k) applies `dst_rank < knn_k` (self + knn_k - 1 neighbours) exactly as
            # it does to the reference's 60-neighbour pools (knn_graph.py:264-266)
            pool = min(MAX_EXACT_K, self.knn_k + 1)
            self._knng = KNNGraph(compute_exact_knn(self.vectors, n_neighbors=pool, device_index=dev,
                                                    device=getattr(self, "device", 0) or 0))
        return self._knng

    def load_index(self, i_name=None, *, options=None):
        if self._index is None:
            tiles = self.vector_meta.groupby("dbidx").size().max()
            if tiles == 1 and i_name != "multiscale":
                from .indices.coarse.coarse_index import CoarseIndex
                idx = CoarseIndex(embedding=self.embedding, vectors=self.vectors, vector_meta=self.vector_meta,
                                  path=self.path, device=self.device)
            else:
                from .indices.multiscale.multiscale_index import MultiscaleIndex
                idx = MultiscaleIndex(embedding=self.embedding, vectors=self.vectors, vector_meta=self.vector_meta,
                                      vec_index=None, path=self.path, device=self.device)
            self._index = idx  # before the graph is built: knn_graph() then uses the matrix the index already holds in HBM
            if self.knn_k > 0:
                graph = self.knn_graph()
                idx.knng = {name: graph for name in ("exact", "nndescent60", "")}
        return self._index


class GlobalDataManager:
    """`gdm.get_dataset(name)` over a dict of SyntheticDataset (seesaw/dataset_manager.py stand-in)."""

    def __init__(self, root=None, datasets: dict = None):
        self.root = root
        self.datasets = dict(datasets or {})

    def add(self, ds: SyntheticDataset):
        self.datasets[ds.name] = ds
        return self

    def list_datasets(self):
        return sorted(self.datasets)

    def get_dataset(self, name):
        return self.datasets[name]


def _tile_boxes(t: int, width=640.0, height=480.0):
    """t boxes: a coarse full-image tile (zoom 1) + a grid of 224-px tiles (zoom 0), in the
    spirit of the reference's pyramid (multiscale_tools.py:16-117: 12 + 1 tiles at 640x480)."""
    boxes = []
    if t == 1:
        return [(0.0, 0.0, 224.0, 224.0, 0)]
    cols = 4
    for j in range(t - 1):
        x1 = (j % cols) * (width - 224.0) / max(cols - 1, 1)
        y1 = (j // cols) * (height - 224.0) / max((t - 2) // cols, 1)
        boxes.append((x1, y1, x1 + 224.0, y1 + 224.0, 0))
    boxes.append((0.0, 0.0, width, height, 1))
    return boxes


def make_dataset(name: str, *, n_images: int, tiles_per_image: int, n_categories: int = 4,
                 positive_frac: float = 0.01, signal: float = 0.55, dim: int = 512, seed: int = 0, knn_k: int = 0,
                 device: int = 0) -> SyntheticDataset:
    rng = np.random.default_rng(seed)
    t = tiles_per_image
    n = n_images * t
    X = rng.standard_normal((n, dim)).astype(np.float32)
    cats = [f"c{i}" for i in range(n_categories)]
    directions = {c: _unit(rng.standard_normal(dim)) for c in cats}
    boxes_one = _tile_boxes(t)
    meta = pd.DataFrame({
        "dbidx": np.repeat(np.arange(n_images), t),
        "zoom_level": np.tile(np.array([b[4] for b in boxes_one], dtype=np.int16), n_images),
        "x1": np.tile(np.array([b[0] for b in boxes_one], dtype=np.float32), n_images),
        "y1": np.tile(np.array([b[1] for b in boxes_one], dtype=np.float32), n_images),
        "x2": np.tile(np.array([b[2] for b in boxes_one], dtype=np.float32), n_images),
        "y2": np.tile(np.array([b[3] for b in boxes_one], dtype=np.float32), n_images),
    })
    recs = []
    n_pos = max(1, int(round(positive_frac * n_images)))
    for c in cats:
        pos_images = rng.choice(n_images, size=n_pos, replace=False)
        for img in pos_images:
            j = int(rng.integers(0, max(1, t - 1)))  # a fine tile carries the object
            row = img * t + j
            X[row] = X[row] / np.linalg.norm(X[row]) * (1 - signal) + directions[c] * signal
            b = boxes_one[j]
            recs.append(dict(dbidx=int(img), category=c, x1=b[0] + 20, y1=b[1] + 20, x2=b[2] - 20, y2=b[3] - 20,
                             im_width=640, im_height=480))
    X = _unit(X)
    box_data = pd.DataFrame(recs)
    return SyntheticDataset(name, vectors=X, vector_meta=meta, box_data=box_data, categories=cats,
                            directions=directions, knn_k=knn_k, device=device)


def make_c1(seed: int = 0, device: int = 0) -> SyntheticDataset:
    """BASELINE.json configs[0]: 10k x 512 random unit vectors, one per image; category c0 =
    the 1 % of images scoring highest against a hidden unit vector."""
    n, dim = 10_000, 512
    X = _unit(np.random.default_rng(seed).standard_normal((n, dim)))
    hidden = _unit(np.random.default_rng(seed + 2).standard_normal(dim))
    pos = np.argsort(-(X @ hidden))[: n // 100]
    meta = pd.DataFrame({"dbidx": np.arange(n)})
    box_data = pd.DataFrame({"dbidx": pos, "category": "c0", "x1": 0.0, "y1": 0.0, "x2": 224.0, "y2": 224.0,
                             "im_width": 224, "im_height": 224})
    query = _unit(np.random.default_rng(seed + 1).standard_normal(dim))
    # the text query: a unit vector correlated with the hidden direction (a plain random query
    # would find nothing in 10k images; the scan cost is identical either way)
    directions = {"c0": _unit(hidden + 0.5 * query)}
    return SyntheticDataset("c1", vectors=X, vector_meta=meta, box_data=box_data, categories=["c0"],
                            directions=directions, device=device,
                            embedding=SyntheticEmbedding(directions, dim, noise=0.0))


def make_lvis_shape(n_images: int = 1109, seed: int = 0, knn_k: int = 10, device: int = 0) -> SyntheticDataset:
    """median LVIS per-category subset: 1 109 images x 13 tiles = 14 417 vectors (BASELINE C5)."""
    return make_dataset("lvis", n_images=n_images, tiles_per_image=13, n_categories=4, positive_frac=0.012,
                        seed=seed, knn_k=knn_k, device=device)
