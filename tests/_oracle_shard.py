"""Test infrastructure: a CPU stand-in for seesaw_amd.indices.multiscale.sharded_index.DeviceShard backed by the
oracle, so that the sharding logic of ShardedMultiscaleIndex (partition, restricted exclusion, exchange, merge,
owner-side stage 2) runs under gloo in the GPU-less container.  Never imported by the product."""
import numpy as np
import torch

from oracle import seesaw_oracle as orc
from seesaw_amd.indices.multiscale.sharded_index import encode_keys

K_BUF = 4096


def merge_on_cpu(device, stream, keys, counts, k, out_keys, out_count):
    """stand-in for ssw_topk_merge_dev: same contract (descending u64 order, each list's first k keys)"""
    lists = [keys[r, : min(int(counts[r]), k)].numpy().view(np.uint64) for r in range(keys.shape[0])]
    allk = np.sort(np.concatenate(lists))[::-1][:k]
    out_keys[: allk.shape[0]] = torch.from_numpy(allk.view(np.int64).copy())
    out_count[0] = allk.shape[0]


class OracleShard:
    def __init__(self, vectors, row2image, boxes, zoom, device):
        self.X, self.r2i, self.boxes, self.zoom = vectors, np.asarray(row2image), boxes, zoom
        self.n_images = int(self.r2i[-1]) + 1
        self.row_start = np.concatenate(([0], np.cumsum(np.bincount(self.r2i))))
        self._scores = None

    def select(self, q, k, excluded_local):
        self._scores = orc.scores_kernel_order(self.X, np.asarray(q, np.float32).reshape(-1))
        ids, sc, rows = orc.topk_images_tiebreak(self._scores, self.r2i, self.n_images, list(excluded_local), k)
        keys = torch.zeros(K_BUF, dtype=torch.int64)
        keys[: ids.shape[0]] = torch.from_numpy(encode_keys(sc, ids).view(np.int64).copy())
        best = torch.zeros(K_BUF, dtype=torch.int64)
        best[: ids.shape[0]] = torch.from_numpy(np.asarray(rows, dtype=np.int64))
        return keys, torch.tensor([ids.shape[0], 0], dtype=torch.int32), best

    select_deep = None

    def select_scores(self, row_scores, k, excluded_local):
        self._scores = np.asarray(row_scores, dtype=np.float32)
        ids, sc, rows = orc.topk_images_tiebreak(self._scores, self.r2i, self.n_images, list(excluded_local), k)
        keys = torch.zeros(K_BUF, dtype=torch.int64)
        keys[: ids.shape[0]] = torch.from_numpy(encode_keys(sc, ids).view(np.int64).copy())
        best = torch.zeros(K_BUF, dtype=torch.int64)
        best[: ids.shape[0]] = torch.from_numpy(np.asarray(rows, dtype=np.int64))
        return keys, torch.tensor([ids.shape[0], 0], dtype=torch.int32), best

    def rows(self, rows_local):
        return self.X[np.asarray(rows_local, dtype=np.int64)]

    def rescore_avg(self, local_positions, aug_larger, minus, aug_weight="level_max"):
        assert aug_weight == "level_max"
        scores, rows, off = [], [], 0
        for p in local_positions:
            a, b = self.row_start[p], self.row_start[p + 1]
            s = self._scores[a:b] if minus is None else (self._scores[a:b] - minus[off:off + (b - a)]).astype(np.float32)
            off += b - a
            j, sc, _ = orc.avg_score_image(self.boxes[a:b], self.zoom[a:b], s, aug_larger)
            scores.append(sc)
            rows.append(a + j)
        return np.asarray(scores, np.float32), np.asarray(rows, np.int64)

    def tile_scores(self, rows_local):
        return self._scores[rows_local]

    def score_rows(self, q2, rows_local):
        return orc.scores_kernel_order(self.X[rows_local], np.asarray(q2, np.float32).reshape(-1))

    def scores(self, q):
        return orc.scores_kernel_order(self.X, np.asarray(q, np.float32).reshape(-1))

    def close(self):
        pass
