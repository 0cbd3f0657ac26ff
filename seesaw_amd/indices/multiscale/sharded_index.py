"""ShardedMultiscaleIndex: the multiscale index row-sharded over the GPUs of one node, behind the same
AccessMethod interface (`query`, `new_query`, `string2vec`, `__len__`).

The reference's index is one numpy array in host RAM (seesaw/indices/multiscale/multiscale_index.py:201-352);
BASELINE.json's north_star shards it: "the vector index shards by rows across the 8 GPUs of one node with an
RCCL all-gather of per-shard top-k".  One process per GPU (torch.distributed); rank r holds the tile vectors of
a contiguous range of images (rows are stored sorted by image, coarse_index.py:49, so the cut is at an image
boundary and an image's tiles never straddle ranks):

  stage 1  every rank: exclusion bitmap restricted to its images, scan + per-image max + exact top-`shortlist`
           on its slice (the kernels of the unsharded index);
           ONE all_gather_into_tensor of [keys | best rows | count+overflow] per rank (seesaw_amd.sharded.ShardedTopK),
           merge of world x shortlist keys on every rank (ssw_topk_merge_dev);
  stage 2  plain_score: the best tile came back with the key -- nothing else to do;
           avg_score / vector2: each candidate image is re-scored on the rank that owns its tiles (their scores are
           resident there), the per-image (score, tile) pairs are all-gathered (a few hundred bytes) and every
           rank finishes with the reference's `np.argsort(-score)[:topk]` (multiscale_index.py:379-403).

Every rank returns the same answer.  The host-side meta (dbidx, boxes, zoom levels: 28 B per tile) is replicated;
the 2 KB per tile of vectors is what is sharded.  `vectors` may be the full host array (the loops that fit on labelled
rows read `index.vectors[rows]`, multi_reg.py:204) or None.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import pandas as pd

from ...bitmap import BitMap
from ...sharded import ShardedTopK, _DevArray, shard_bounds_by_image
from ..interface import ActivationFrames
from .multiscale_index import MultiscaleIndex, _Candidates


def encode_keys(scores: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """composite keys as the select kernels emit them: (orderable(score) << 32) | (0xFFFFFFFF - id)"""
    u = np.ascontiguousarray(scores, dtype=np.float32).view(np.uint32).astype(np.uint64)
    neg = (u & np.uint64(0x80000000)) != 0
    o = np.where(neg, ~u & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    return (o << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.asarray(ids, dtype=np.uint64))


class DeviceShard:
    """one rank's slice in HBM: the same DeviceIndex the unsharded index uses, results left on the device"""

    def __init__(self, vectors: np.ndarray, row2image: np.ndarray, boxes: Optional[np.ndarray], zoom: Optional[np.ndarray],
                 device: int):
        import torch
        from ... import _lib
        from ...device_index import DeviceIndex
        self.torch = torch
        self.tdev = torch.device("cuda", device)
        self.index = DeviceIndex.from_numpy(vectors, row2image=row2image.astype(np.int32), device=device)
        if boxes is not None:
            self.index.set_tile_meta(boxes, zoom)
        self.index.set_stream(torch.cuda.current_stream(self.tdev).cuda_stream)
        keys_ptr, count_ptr, best_ptr = self.index.result_ptrs()
        self.keys = torch.as_tensor(_DevArray(keys_ptr, (_lib.SSW_MAX_TOPK,), "<i8"), device=self.tdev)
        self.count = torch.as_tensor(_DevArray(count_ptr, (2,), "<i4"), device=self.tdev)
        # u32 row numbers viewed as i32 (torch's uint32 support is partial); widened with a mask where they are read
        self.best = torch.as_tensor(_DevArray(best_ptr, (_lib.SSW_MAX_TOPK,), "<i4"), device=self.tdev)

    def select(self, q: np.ndarray, k: int, excluded_local: np.ndarray):
        """scan + top-k of the slice; -> (keys int64 [>=k], (count, overflow) int32 [2], best local rows int64 [>=k])"""
        qd = self.torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1)).to(self.tdev)
        self.index.set_excluded(excluded_local)
        self.index.topk_dev(qd.data_ptr(), k)
        self._q_keepalive = qd
        return self.keys, self.count, self.best.to(self.torch.int64) & 0xFFFFFFFF

    def select_deep(self, k: int):
        self.index.select_deep_dev(k)
        return self.keys, self.count, self.best.to(self.torch.int64) & 0xFFFFFFFF

    def select_scores(self, row_scores: np.ndarray, k: int, excluded_local: np.ndarray):
        """top-k of the slice by caller-supplied per-row scores (f32, -inf = skip): the label-propagation ranking"""
        self.index.set_excluded(excluded_local)
        self.index.load_scores(row_scores)
        self.index.topk_dev(0, k)
        return self.keys, self.count, self.best.to(self.torch.int64) & 0xFFFFFFFF

    def rows(self, rows_local):
        return self.index.gather_rows(rows_local)

    def rescore_avg(self, local_positions, aug_larger, minus, aug_weight="level_max"):
        return self.index.rescore_avg(local_positions, aug_larger, minus, aug_weight=aug_weight)

    def tile_scores(self, rows_local):
        return self.index.gather_scores(rows_local)

    def score_rows(self, q2, rows_local):
        return self.index.score_rows(q2, rows_local)

    def scores(self, q):
        return self.index.scores(q)

    def close(self):
        self.index.close()


class _ShardedRows:
    """`index.vectors` of a sharded index built without the full host array: what the loops read of it --
    `.shape`, `vectors[rows]` (multi_reg.py:204, loops/util.py:6,11, rocchio_update.py:24) -- served from the shards.
    `vectors[rows]` is a COLLECTIVE (every rank runs the same session and asks for the same rows)."""

    def __init__(self, index, dim):
        self._index, self.shape, self.dtype, self.ndim = index, (index.vector_meta.shape[0], dim), np.dtype(np.float32), 2

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        """the whole matrix on this rank (COLLECTIVE; N x 2 KB): only the one-off X'LX of the data regulariser asks
        for it (loops/graph_based.py compute_xlx), which every rank then forms redundantly on its own GPU"""
        out = self._index.gather_rows(np.arange(self.shape[0], dtype=np.int64))
        return out if dtype is None else out.astype(dtype, copy=False)

    def __getitem__(self, rows):
        if isinstance(rows, tuple):
            raise TypeError("a sharded index serves whole rows: vectors[rows]")
        if isinstance(rows, slice):
            rows = np.arange(*rows.indices(self.shape[0]))
        rows = np.asarray(rows)
        if rows.dtype == bool:
            rows = np.flatnonzero(rows)
        one = rows.ndim == 0
        out = self._index.gather_rows(rows.reshape(-1).astype(np.int64))
        return out[0] if one else out


class ShardedMultiscaleIndex(MultiscaleIndex):
    def __init__(self, *, embedding, vectors: Optional[np.ndarray], vector_meta: pd.DataFrame, rank: int, world: int,
                 local_vectors: Optional[np.ndarray] = None, device: int = 0, group=None, k_max: int = 1024,
                 comm_device=None, shard_factory=DeviceShard, merge=None, path: str = None, excluded: BitMap = None):
        """vectors: the full [N, 512] host array (sliced here) or None with `local_vectors` = this rank's rows.
        comm_device: where the collective's tensors live -- None = the shard's GPU (backend nccl = RCCL); "cpu"
        for a gloo group (the messages make a host round trip; used by the tests on one-GPU boxes)."""
        self.rank, self.world, self.group = int(rank), int(world), group
        self._k_max, self._comm_device, self._shard_factory, self._merge = int(k_max), comm_device, shard_factory, merge
        self._local_vectors = local_vectors
        serve_rows = vectors is None
        if vectors is None:
            assert local_vectors is not None
            vectors = np.zeros((vector_meta.shape[0], 0), dtype=np.float32)  # placeholder: meta only
        super().__init__(embedding=embedding, vectors=vectors, vector_meta=vector_meta, vec_index=None, path=path,
                         excluded=excluded, device=device)
        if serve_rows:  # no rank holds the matrix: `index.vectors[rows]` gathers from the owning shards
            self.vectors = _ShardedRows(self, int(local_vectors.shape[1]))

    @staticmethod
    def row_range(vector_meta: pd.DataFrame, world: int, rank: int):
        """[row_lo, row_hi) of `rank`: what a loader reads from vectors.sorted.cached for this rank"""
        counts = np.bincount(np.unique(vector_meta.dbidx.values, return_inverse=True)[1])
        row_start = np.concatenate(([0], np.cumsum(counts)))
        _, _, lo, hi = shard_bounds_by_image(row_start, world, rank)
        return lo, hi

    # ---- construction -------------------------------------------------------------------
    def _init_device(self):
        import torch
        self.img_lo, self.img_hi, self.row_lo, self.row_hi = shard_bounds_by_image(self._row_start, self.world, self.rank)
        local = self._local_vectors if self._local_vectors is not None else self.vectors[self.row_lo:self.row_hi]
        assert local.shape[0] == self.row_hi - self.row_lo, "local_vectors must hold exactly this rank's rows"
        self.n_local_images = self.img_hi - self.img_lo
        self._shard = None
        if self.n_local_images > 0:
            r2i = self._row2pos[self.row_lo:self.row_hi] - self.img_lo
            boxes = self._box[self.row_lo:self.row_hi] if self._has_tile_meta else None
            zoom = self.vector_meta.zoom_level.values[self.row_lo:self.row_hi] if self._has_tile_meta else None
            self._shard = self._shard_factory(np.ascontiguousarray(local, dtype=np.float32), r2i, boxes, zoom, self.device)
        compute_dev = torch.device("cuda", self.device) if self._shard_factory is DeviceShard else torch.device("cpu")
        kw = {} if self._merge is None else {"merge": self._merge}
        self._xchg = ShardedTopK(rank=self.rank, world=self.world, device=compute_dev, image_offset=self.img_lo,
                                 k_max=self._k_max, group=self.group, with_best=True, comm_device=self._comm_device, **kw)
        self._dev = None  # there is no whole-matrix device index

    # ---- stage 1 ------------------------------------------------------------------------
    def _prelim(self, *, vector, topk_dbidx, exclude_dbidx=None, force_exact=False):
        cand = self._select_and_exchange(lambda shard, k_local, mine: shard.select(vector, k_local, mine), topk_dbidx,
                                         exclude_dbidx)
        if not isinstance(cand, tuple):
            self._resident_q = np.asarray(vector, dtype=np.float32).reshape(-1).copy()
        return cand

    def _select_and_exchange(self, select, topk_dbidx, exclude_dbidx, keep_finite=False):
        """local selection (select(shard, k_local, excluded_local) -> keys, (count, overflow), best local rows),
        one all-gather of the messages, merge on every rank -> _Candidates (or the empty triple)"""
        excl_pos = self._excluded_positions(exclude_dbidx)
        n_included = self._dbidx.shape[0] - excl_pos.shape[0]
        k = min(int(topk_dbidx), n_included)
        if k == 0:
            print("no dbidx included")
            return [], [], []
        assert k <= self._k_max, f"shortlist {k} exceeds the exchange buffers (k_max={self._k_max})"
        mine = excl_pos[(excl_pos >= self.img_lo) & (excl_pos < self.img_hi)] - self.img_lo
        x = self._xchg
        if self._shard is not None:
            k_local = min(k, self.n_local_images)
            keys, count, best = select(self._shard, k_local, mine)
            x.pack(keys, count, k_local, best_rows=best + self.row_lo)
        else:
            x.pack_empty()
        x.gather()
        out_keys, out_count = x.merge_gathered(k)
        over = x.overflowed()   # host read: synchronises
        if over:                # same list on every rank: the flagged ones redo their selection exactly, all re-exchange
            if self.rank in over:
                keys, count, best = self._shard.select_deep(min(k, self.n_local_images))
                x.pack(keys, count, min(k, self.n_local_images), best_rows=best + self.row_lo)
            x.gather()
            out_keys, out_count = x.merge_gathered(k)
            assert not x.overflowed()
            x.reset_overflow_seen()
        from ...device_index import decode_keys
        c = int(out_count.cpu().item())
        merged = out_keys[:c].cpu().numpy().view(np.uint64)
        pos, scores = decode_keys(merged)
        best_rows = x.best_rows_of(merged)
        if keep_finite:  # images whose every row was skipped (score -inf) do not take part
            keep = np.isfinite(scores)
            pos, scores, best_rows = pos[keep], scores[keep], best_rows[keep]
        return _Candidates(self._dbidx[pos], scores, pos, best_rows)

    def topk_from_scores(self, row_scores, *, topk_dbidx, exclude_dbidx=None, skip_rows=None):
        """the selection by caller-supplied per-row scores (the label-propagation ranking, graph_based.py:88-98): every rank
        holds the whole score vector (the propagation is replicated), loads its slice and takes part in the usual exchange"""
        s = np.asarray(row_scores, dtype=np.float32).copy()
        if skip_rows is not None:
            s[skip_rows] = -np.inf
        local = s[self.row_lo:self.row_hi]
        self._resident_q = None
        cand = self._select_and_exchange(lambda shard, k_local, mine: shard.select_scores(local, k_local, mine), topk_dbidx,
                                         exclude_dbidx, keep_finite=True)
        if isinstance(cand, tuple):
            return _Candidates(np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.int64), np.zeros(0, np.int64))
        return cand

    topk_from_device_scores = None  # (the propagated scores of a graph loop reach a sharded index through the host)

    # ---- stage 2 ------------------------------------------------------------------------
    def _owned(self, positions: np.ndarray) -> np.ndarray:
        return positions[(positions >= self.img_lo) & (positions < self.img_hi)]

    def _comm_tensor_device(self):
        import torch
        if self._comm_device is not None:
            return torch.device(self._comm_device)
        return torch.device("cuda", self.device) if self._shard_factory is DeviceShard else torch.device("cpu")

    def _all_gather_f64(self, vec: np.ndarray, cap: int):
        """fixed-size tensor collective: every rank contributes <= cap float64 values -> list of per-rank arrays.
        (One all_gather_into_tensor of world x (cap + 1) doubles; under RCCL an object collective would pickle, stage
        through byte tensors and run two collectives.)"""
        import torch
        import torch.distributed as dist
        n = int(vec.shape[0])
        assert n <= cap, (n, cap)
        msg = np.zeros(cap + 1, dtype=np.float64)
        msg[0] = n
        msg[1:1 + n] = vec
        dev = self._comm_tensor_device()
        send = torch.from_numpy(msg).to(dev)
        recv = torch.empty(self.world * (cap + 1), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        out = recv.cpu().numpy().reshape(self.world, cap + 1)
        return [out[r, 1:1 + int(out[r, 0])] for r in range(self.world)]

    def _gather_stage2(self, own_pos, own_scores, own_rows, all_positions):
        """(score, tile row) of every candidate image, computed on its owner, known to every rank afterwards"""
        mine = (np.asarray(own_pos, np.int64), np.asarray(own_scores, np.float32), np.asarray(own_rows, np.int64))
        if self.world > 1:
            # positions, scores (f32) and rows (< 2^53) are exact in float64: one fixed-size message per rank
            cap = int(all_positions.shape[0])
            flat = np.concatenate([mine[0].astype(np.float64), mine[1].astype(np.float64), mine[2].astype(np.float64)])
            parts = []
            for v in self._all_gather_f64(flat, 3 * cap):
                m = v.shape[0] // 3
                parts.append((v[:m].astype(np.int64), v[m:2 * m].astype(np.float32), v[2 * m:].astype(np.int64)))
        else:
            parts = [mine]
        pos = np.concatenate([p[0] for p in parts])
        order = np.argsort(pos, kind="stable")
        assert np.array_equal(pos[order], all_positions), "every candidate image has exactly one owner"
        return np.concatenate([p[1] for p in parts])[order], np.concatenate([p[2] for p in parts])[order]

    def _rescore_avg_on_device(self, candidate_df, topk, aug_larger, vector2=None, aug_weight="level_max"):
        positions = np.sort(np.asarray(candidate_df.attrs["positions"], dtype=np.int64))
        own = self._owned(positions)
        scores, rows = np.zeros(0, np.float32), np.zeros(0, np.int64)
        if own.size:
            minus = None
            if vector2 is not None:
                minus = self._shard.score_rows(vector2, self._candidate_rows(own) - self.row_lo)
            scores, rows = self._shard.rescore_avg(own - self.img_lo, aug_larger, minus, aug_weight)
            rows = rows + self.row_lo
        scores, rows = self._gather_stage2(own, scores, rows, positions)
        top = np.argsort(-scores.astype(np.float64))[:topk]
        return self._activations(positions[top], rows[top], scores[top])

    def _plain_vector2(self, candidate_df, topk, vector2):
        """plain_score with a second vector: per candidate image the first tile maximising s1 - s2"""
        positions = np.sort(np.asarray(candidate_df.attrs["positions"], dtype=np.int64))
        own = self._owned(positions)
        scores, rows = [], []
        if own.size:
            ilocs = self._candidate_rows(own)
            s = self._shard.tile_scores(ilocs - self.row_lo) - self._shard.score_rows(vector2, ilocs - self.row_lo)
            bounds = np.concatenate(([0], np.cumsum(self._row_start[own + 1] - self._row_start[own])))
            for a, b in zip(bounds[:-1], bounds[1:]):
                j = int(np.argmax(s[a:b]))  # first maximum
                scores.append(s[a + j])
                rows.append(ilocs[a + j])
        scores, rows = self._gather_stage2(own, np.asarray(scores, np.float32), np.asarray(rows, np.int64), positions)
        top = np.argsort(-scores.astype(np.float64), kind="stable")[:topk]
        return self._activations(positions[top], rows[top], scores[top])

    def _activations(self, positions, rows, scores):
        return {"dbidxs": self._dbidx[positions].astype("int"),
                "activations": ActivationFrames(self._box[rows], self._row_dbidx[rows], np.asarray(scores))}

    def query(self, *, vector, vector2=None, topk, shortlist_size, exclude=None, force_exact=False, **kwargs):
        if shortlist_size is None:
            shortlist_size = topk * 5
        candidate_df = self._prelim(vector=vector, topk_dbidx=shortlist_size, exclude_dbidx=exclude,
                                    force_exact=force_exact)
        if isinstance(candidate_df, tuple):
            return {"dbidxs": np.zeros(0, dtype="int"), "activations": []}
        agg_method = kwargs.get("agg_method")
        if agg_method == "plain_score":
            if vector2 is None:
                return self._activations_from_best(candidate_df, topk)
            return self._plain_vector2(candidate_df, topk, vector2)
        if not self._has_tile_meta:
            raise NotImplementedError("the sharded index aggregates over float32 tile boxes with zoom levels <= 31")
        return self._rescore_avg_on_device(candidate_df, topk, kwargs["aug_larger"], vector2,
                                           aug_weight=kwargs.get("aug_weight", "level_max"))

    # ---- the rest of the interface --------------------------------------------------------
    def score(self, vec):
        """all N scores on every rank: each computes its slice, the slices are all-gathered (label propagation and
        calibration read the full vector; N x 4 bytes)"""
        mine = self._shard.scores(vec) if self._shard is not None else np.zeros(0, np.float32)
        if self.world == 1:
            return mine
        import torch
        import torch.distributed as dist
        # every rank's row range is known everywhere (shard_bounds_by_image): one padded tensor collective
        bounds = [shard_bounds_by_image(self._row_start, self.world, r) for r in range(self.world)]
        cap = max(b[3] - b[2] for b in bounds)
        dev = self._comm_tensor_device()
        send = torch.zeros(cap, dtype=torch.float32, device=dev)
        send[:mine.shape[0]] = torch.from_numpy(np.ascontiguousarray(mine, dtype=np.float32)).to(dev)
        recv = torch.empty(self.world * cap, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        out = recv.cpu().numpy().reshape(self.world, cap)
        return np.concatenate([out[r, :bounds[r][3] - bounds[r][2]] for r in range(self.world)])

    def gather_rows(self, rows) -> np.ndarray:
        """`vectors[rows]` [m, dim] on every rank without a host copy of the matrix anywhere: each rank reads the rows it
        owns out of its shard (ssw_index_gather_rows), the others' places stay zero, one all-reduce(sum) fills them in
        (x + 0 + ... + 0 is exact).  COLLECTIVE: every rank calls it with the same rows."""
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        dim = self.vectors.shape[1]
        out = np.zeros((rows.shape[0], dim), dtype=np.float32)
        own = (rows >= self.row_lo) & (rows < self.row_hi)
        if own.any() and self._shard is not None:
            out[own] = self._shard.rows(rows[own] - self.row_lo)
        if self.world > 1 and rows.shape[0]:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(out).to(self._comm_tensor_device())
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            out = t.cpu().numpy()
        return out

    def subset(self, indices):
        raise NotImplementedError("subset of a sharded index")

    def close(self):
        if self._shard is not None:
            self._shard.close()
