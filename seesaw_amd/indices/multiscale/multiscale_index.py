"""MultiscaleIndex: many tile vectors per image, two-stage lookup on the GPU.

Interface of seesaw/indices/multiscale/multiscale_index.py:201-442 (MultiscaleIndex,
BoxFeedbackQuery, rescore_candidates, match_labels_to_vectors).

stage 1 (reference :291-312 = `vectors @ q`, full `np.argsort`, pandas gather / isin /
np.unique over all N rows): one HBM-streaming scan + fused per-image max + exact radix
select of the top `shortlist_size` distinct non-excluded images (ssw_index_topk).
stage 2 (reference :341-352, 379-403): the candidate images' tile scores are read back
from the score buffer the scan left in HBM (ssw_index_gather_scores) -- no second matvec,
no `isin` over N rows -- and aggregated per image on the host (<= shortlist_size groups).
"""
from __future__ import annotations

import json
import os

import numpy as np
import pandas as pd

from ...bitmap import BitMap, FrozenBitMap
from ...box_utils import box_iou, left_iou_join
from ...device_index import DeviceIndex
from ...labeldb import LabelDB
from ...query_interface import AccessMethod, InteractiveQuery
from ..coarse.coarse_index import _positions_of
from ..interface import ActivationFrames, resolve_path

_ACT_COLS = ["x1", "y1", "x2", "y2", "dbidx", "score"]


def distinct_topk_positions(dbidxs, topk):
    """positions of the first occurrence of the first `topk` distinct values
    (same contract as multiscale_index.py:177-180; kept for callers and tests)."""
    _, first = np.unique(dbidxs, return_index=True)
    return np.sort(first)[:topk]


def match_labels_to_vectors(label_db: LabelDB, vec_meta: pd.DataFrame, target_description=None):
    """tile rows of every seen image with ys = 1 iff the tile overlaps an accepted label box
    (multiscale_index.py:65-83)."""
    seen = np.asarray(label_db.get_seen(), dtype=np.int64)
    vec_meta = vec_meta[np.isin(vec_meta.dbidx.values, seen)]
    boxdf = label_db.get_box_df(return_description=True)
    if target_description is not None:
        target_df = boxdf[boxdf.description == target_description]
    else:
        target_df = boxdf[boxdf.marked_accepted > 0]
    out = left_iou_join(vec_meta, target_df)
    return out.assign(ys=(out.max_iou > 0).astype("float"))


def _kahan_mean_rows(vals: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """per row: mean of vals where mask, summed column by column with a Kahan-compensated sum in vals' dtype --
    pandas' group_mean, which is what the reference's `.groupby('iloc_left').score_right.mean()` runs"""
    dt = vals.dtype
    total = np.zeros(vals.shape[0], dtype=dt)
    comp = np.zeros(vals.shape[0], dtype=dt)
    count = mask.sum(axis=1)
    for col in range(vals.shape[1]):
        m = mask[:, col]
        y = np.where(m, vals[:, col] - comp, dt.type(0))
        t = total + y
        comp = np.where(m, (t - total) - y, comp)
        total = np.where(m, t, total)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(count > 0, total / count.astype(dt), dt.type(np.nan))


def _avg_score_per_tile(meta_df: pd.DataFrame, aug_larger: str, aug_weight: str = "level_max"):
    """Host form of the `avg_score` aggregation for the tiles of ONE image, as [T, T] matrices (the product path is the
    device kernel, csrc/rescore.hip; this serves indexes whose tile geometry is not float32 / zoom level > 31).
    Semantics of score_frame2 (multiscale_index.py:112-150): tile i is joined with every tile j of IoU > 0 whose zoom
    level passes `aug_larger`; 'level_max': mean over the partner levels of the score of the first best-overlapping
    partner of that level; 'cont_weighted': softmax(containment) weights over all partners."""
    iou, cont = box_iou(meta_df, meta_df, return_containment=True)
    zoom = meta_df.zoom_level.to_numpy()
    score = meta_df.score.to_numpy()
    joined = iou > 0
    if aug_larger == "greater":
        joined &= zoom[None, :] >= zoom[:, None]
    elif aug_larger == "adjacent":
        joined &= zoom[None, :] == zoom[:, None]
    else:
        assert aug_larger == "all", aug_larger
    T = zoom.shape[0]
    if aug_weight == "cont_weighted":
        from scipy.special import softmax
        out = np.full(T, np.nan, dtype=score.dtype)
        for i in np.flatnonzero(joined.any(axis=1)):
            out[i] = softmax(cont[i, joined[i]]) @ score[joined[i]]
        return pd.Series(out)
    assert aug_weight == "level_max", aug_weight
    levels = np.unique(zoom)  # ascending: the order pandas sums the groups in
    picked = np.zeros((T, levels.shape[0]), dtype=score.dtype)
    has = np.zeros((T, levels.shape[0]), dtype=bool)
    for k, z in enumerate(levels):
        cols = np.flatnonzero(zoom == z)
        sub = np.where(joined[:, cols], iou[:, cols], -1.0)
        best = sub.argmax(axis=1)  # first maximum, as idxmax
        has[:, k] = sub[np.arange(T), best] > 0
        picked[:, k] = score[cols[best]]
    return pd.Series(_kahan_mean_rows(picked, has))


def score_frame2(meta_df: pd.DataFrame, **aug_options):
    """best tile of one image under the requested aggregation; one-row frame."""
    agg_method = aug_options["agg_method"]
    if agg_method != "plain_score":
        meta_df = meta_df.reset_index(drop=True)
        scores = _avg_score_per_tile(meta_df, aug_options["aug_larger"], aug_options.get("aug_weight", "level_max"))
        meta_df = meta_df.assign(unadjusted_score=meta_df.score, score=scores)
    return meta_df[meta_df.score == meta_df.score.max()].head(n=1)


def rescore_candidates(fullmeta: pd.DataFrame, topk: int, **kwargs):
    """per candidate image: aggregate tile scores, keep the best tile as its activation;
    return the `topk` best images (multiscale_index.py:379-403)."""
    fullmeta = fullmeta.reset_index(drop=True)
    dbidxs, dbscores, activations = [], [], []
    plain = kwargs.get("agg_method") == "plain_score"
    if plain and fullmeta.shape[0]:
        # vectorised form of the per-frame loop: first maximal row of every image; only the
        # `topk` winners get an activation frame built
        d = fullmeta.dbidx.values
        s = fullmeta.score.values
        order = np.lexsort((np.arange(d.shape[0]), -s, d))
        firsts = order[np.concatenate(([True], d[order][1:] != d[order][:-1]))]
        top = firsts[np.argsort(-s[firsts].astype(np.float64), kind="stable")[:topk]]
        return {"dbidxs": d[top].astype("int"), "activations": [fullmeta.iloc[[i]][_ACT_COLS] for i in top]}
    else:
        for dbidx, frame_meta in fullmeta.groupby("dbidx"):
            tup = score_frame2(frame_meta, **kwargs)
            dbidxs.append(dbidx)
            dbscores.append(tup.score.iloc[0])
            activations.append(tup[_ACT_COLS])
    dbidxs = np.asarray(dbidxs)
    dbscores = np.asarray(dbscores, dtype=np.float64)
    top = np.argsort(-dbscores, kind="stable")[:topk]
    return {"dbidxs": dbidxs[top].astype("int"), "activations": [activations[i] for i in top]}


class _Candidates:
    """what _query_prelim hands to the second stage: the reference's two-column frame (dbidx, max_score) as plain
    arrays plus the image positions and best rows that came back with the selection; `.df` builds the DataFrame
    for callers that want the reference's return type"""
    __slots__ = ("dbidx", "max_score", "attrs", "_df")

    def __init__(self, dbidx, max_score, positions, best_rows):
        self.dbidx, self.max_score = dbidx, max_score
        self.attrs = {"positions": positions, "best_rows": best_rows}
        self._df = None

    @property
    def df(self) -> pd.DataFrame:
        if self._df is None:
            self._df = pd.DataFrame({"dbidx": self.dbidx, "max_score": self.max_score})
            self._df.attrs.update(self.attrs)
        return self._df

    def __len__(self):
        return len(self.dbidx)


class MultiscaleIndex(AccessMethod):
    """implements a two stage lookup"""

    def __init__(self, *, embedding, vectors: np.ndarray, vector_meta: pd.DataFrame, vec_index=None,
                 min_zoom_level=1, path: str = None, excluded: BitMap = None, device: int = 0):
        self.embedding = embedding
        self.path = path
        self.excluded = BitMap([]) if excluded is None else excluded
        if min_zoom_level != 1:  # drop the finest zoom level except where it is the only one
            print("WARNING: filtering out min_zoom_level")
            zmax = vector_meta.groupby("dbidx").zoom_level.transform("max")
            keep = ((vector_meta.zoom_level == zmax) | (vector_meta.zoom_level >= min_zoom_level)).values
            vector_meta = vector_meta[keep].reset_index(drop=True)
            vectors = vectors[keep]
        self.vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        self.vector_meta = vector_meta.reset_index(drop=True)
        self.vec_index = vec_index  # accepted for signature parity; the exact GPU scan is always used
        row_dbidx = np.asarray(self.vector_meta.dbidx.values, dtype=np.int64)
        assert np.all(np.diff(row_dbidx) >= 0), "rows must be sorted by dbidx (vectors.sorted.cached)"
        self._row_dbidx = row_dbidx
        self._box = np.stack([self.vector_meta[c].values for c in ("x1", "y1", "x2", "y2")], axis=1) \
            if all(c in self.vector_meta for c in ("x1", "y1", "x2", "y2")) else np.zeros((row_dbidx.shape[0], 4), np.float32)
        self._dbidx, self._row2pos = np.unique(row_dbidx, return_inverse=True)
        self._row_start = np.concatenate(([0], np.cumsum(np.bincount(self._row2pos))))
        self.all_indices = FrozenBitMap(self._dbidx) - self.excluded
        self.device = device
        self._has_tile_meta = "zoom_level" in self.vector_meta and int(self.vector_meta.zoom_level.max()) <= 31 \
            and int(self.vector_meta.zoom_level.min()) >= 0 and self._box.dtype == np.float32
        self._resident_q = None
        self._init_device()

    def _init_device(self):
        """the whole matrix into HBM (a sharded index overrides this with its own slice)"""
        self._dev = DeviceIndex.from_numpy(self.vectors, row2image=self._row2pos.astype(np.int32), device=self.device)
        if self._has_tile_meta:  # tile geometry next to the vectors: the avg_score aggregation runs on the device
            self._dev.set_tile_meta(self._box, self.vector_meta.zoom_level.values)

    # ---- construction -----------------------------------------------------------------
    @staticmethod
    def from_path(index_path: str, *, use_vec_index=True, exclude=None, device: int = 0, **options):
        """<index>/info.json {"constructor", "model", ...} plus either the reference's
        <index>/vectors.sorted.cached parquet (as create_multiscale_index writes it) or
        <index>/vectors.npy [N,512] f32 + <index>/vector_meta.parquet; rows sorted by dbidx."""
        index_path = resolve_path(index_path)
        info = json.load(open(f"{index_path}/info.json"))
        from ...models.embeddings import load_embedding
        embedding = load_embedding(info.get("model"), device=device)
        if os.path.exists(f"{index_path}/vectors.sorted.cached"):  # the reference's on-disk layout (:242-260)
            from .multiscale_tools import read_vector_parquet
            meta, vectors = read_vector_parquet(f"{index_path}/vectors.sorted.cached")
        else:
            vectors = np.load(f"{index_path}/vectors.npy", mmap_mode="r")
            meta = pd.read_parquet(f"{index_path}/vector_meta.parquet").reset_index(drop=True)
        meta = meta[["dbidx", "zoom_level", "x1", "y1", "x2", "y2"]]
        return MultiscaleIndex(embedding=embedding, vectors=np.asarray(vectors), vector_meta=meta,
                               vec_index=None, path=index_path,
                               excluded=info.get("excluded", None) if exclude is None else exclude, device=device)

    def get_knng(self, path=None):
        from ...knn_graph import KNNGraph
        return KNNGraph.from_file(f"{self.path}/knn_graph/{path or ''}")

    def string2vec(self, string: str):
        vec = self.embedding.from_string(string=string)
        return vec / np.linalg.norm(vec)

    def score(self, vec):
        self._resident_q = None
        return self._dev.scores(vec)

    def __len__(self):
        return len(self.all_indices)

    # ---- queries ----------------------------------------------------------------------
    def _excluded_positions(self, exclude_dbidx) -> np.ndarray:
        if exclude_dbidx is not None and len(self.excluded) == 0 and hasattr(exclude_dbidx, "_v"):
            return _positions_of(self._dbidx, exclude_dbidx._v)  # a BitMap's array is sorted and distinct already
        ids = [np.asarray(self.excluded, dtype=np.int64)]
        if exclude_dbidx is not None:
            ids.append(np.asarray(exclude_dbidx, dtype=np.int64))
        return _positions_of(self._dbidx, np.unique(np.concatenate(ids)))

    def _query_prelim(self, *, vector, topk_dbidx, exclude_dbidx=None, force_exact=False):
        """top `topk_dbidx` distinct non-excluded images by their best tile: DataFrame
        (dbidx, max_score) in descending score order (multiscale_index.py:291-312)."""
        cand = self._prelim(vector=vector, topk_dbidx=topk_dbidx, exclude_dbidx=exclude_dbidx, force_exact=force_exact)
        return cand if isinstance(cand, tuple) else cand.df

    def _prelim(self, *, vector, topk_dbidx, exclude_dbidx=None, force_exact=False):
        """_query_prelim without the DataFrame (query() consumes the arrays directly)"""
        excl_pos = self._excluded_positions(exclude_dbidx)
        n_included = self._dbidx.shape[0] - excl_pos.shape[0]
        topk_dbidx = min(int(topk_dbidx), n_included)
        if topk_dbidx == 0:
            print("no dbidx included")
            return [], [], []
        pos, scores, best_rows = self._dev.topk(vector, topk_dbidx, excluded=excl_pos)
        self._resident_q = np.asarray(vector, dtype=np.float32).reshape(-1).copy()
        return _Candidates(self._dbidx[pos], scores, pos, best_rows)

    def topk_from_scores(self, row_scores: np.ndarray, *, topk_dbidx, exclude_dbidx=None, skip_rows=None):
        """same selection as _query_prelim but ranking by caller-supplied per-row scores
        (label-propagation output); `skip_rows` (bool mask) drops rows, e.g. labelled vectors."""
        s = np.asarray(row_scores, dtype=np.float32).copy()
        if skip_rows is not None:
            s[skip_rows] = -np.inf
        excl_pos = self._excluded_positions(exclude_dbidx)
        self._dev.load_scores(s)
        self._resident_q = None
        pos, scores, best_rows = self._dev.topk(None, max(1, min(int(topk_dbidx), self._dbidx.shape[0])), excluded=excl_pos)
        keep = np.isfinite(scores)  # images whose every row was skipped do not take part
        df = pd.DataFrame({"dbidx": self._dbidx[pos[keep]], "max_score": scores[keep]})
        df.attrs["positions"] = pos[keep]
        df.attrs["best_rows"] = best_rows[keep]
        return df

    def topk_from_device_scores(self, fill_scores, *, topk_dbidx, exclude_dbidx=None):
        """topk_from_scores for scores that are already on the device: `fill_scores(device_index)` writes the
        per-row f32 scores (skipped rows at -inf) into the index's score buffer."""
        excl_pos = self._excluded_positions(exclude_dbidx)
        fill_scores(self._dev)
        self._resident_q = None
        pos, scores, best_rows = self._dev.topk(None, max(1, min(int(topk_dbidx), self._dbidx.shape[0])), excluded=excl_pos)
        keep = np.isfinite(scores)
        return _Candidates(self._dbidx[pos[keep]], scores[keep], pos[keep], best_rows[keep])  # (`.df` for a frame)

    def topk_after_update(self, model, idxs, labels, *, topk_dbidx, exclude_dbidx=None):
        """model.update(idxs, labels) + topk_from_device_scores(model.lp.scores_to_index ...) as one device call
        (LabelPropagationRanker2.update_and_select): the shortlist the next next_batch() of a graph loop asks for"""
        excl_pos = self._excluded_positions(exclude_dbidx)
        self._resident_q = None
        pos, scores, best_rows = model.update_and_select(idxs, labels, self._dev, excluded=excl_pos,
                                                         k=max(1, min(int(topk_dbidx), self._dbidx.shape[0])))
        keep = np.isfinite(scores)
        return _Candidates(self._dbidx[pos[keep]], scores[keep], pos[keep], best_rows[keep])

    def _activations_from_best(self, candidate_df: pd.DataFrame, topk: int):
        rows = np.asarray(candidate_df.attrs["best_rows"][:topk], dtype=np.int64)
        scores = np.asarray(candidate_df.max_score)[:topk]
        return {"dbidxs": np.asarray(candidate_df.dbidx)[:topk].astype("int"),
                "activations": ActivationFrames(self._box[rows], self._row_dbidx[rows], scores)}

    def _candidate_rows(self, positions: np.ndarray) -> np.ndarray:
        positions = np.sort(np.asarray(positions, dtype=np.int64))
        return np.concatenate([np.arange(self._row_start[p], self._row_start[p + 1]) for p in positions]) \
            if positions.size else np.zeros(0, dtype=np.int64)

    def query(self, *, vector, vector2=None, topk, shortlist_size, exclude=None, force_exact=False, **kwargs):
        if shortlist_size is None:
            shortlist_size = topk * 5
        if shortlist_size < topk * 5:
            print(f"Warning: shortlist_size parameter {shortlist_size} is small compared to topk param {topk}, "
                  "you may consider increasing it")
        candidate_df = self._prelim(vector=vector, topk_dbidx=shortlist_size, exclude_dbidx=exclude,
                                    force_exact=force_exact)
        if isinstance(candidate_df, tuple):  # nothing left to return
            return {"dbidxs": np.zeros(0, dtype="int"), "activations": []}
        if vector2 is None and kwargs.get("agg_method") == "plain_score":
            # the per-image max and the tile attaining it came back with the selection, and the
            # shortlist is already in (score desc, dbidx asc) order -- the order rescore_candidates
            # would produce -- so the stage-2 gather is not needed
            return self._activations_from_best(candidate_df, topk)
        agg_method = kwargs.get("agg_method")
        if agg_method != "plain_score" and self._has_tile_meta:
            # score_frame2 takes every agg_method other than plain_score down its averaging branch
            # (multiscale_index.py:112-150: 'avg_vector' included): ssw_index_rescore_avg over the resident scores
            return self._rescore_avg_on_device(candidate_df, topk, kwargs["aug_larger"], vector2,
                                               aug_weight=kwargs.get("aug_weight", "level_max"))
        ilocs = self._candidate_rows(candidate_df.attrs["positions"])
        scores = self._dev.gather_scores(ilocs)  # tile scores of the scan that just ran
        if vector2 is not None:
            scores = scores - self._dev.score_rows(vector2, ilocs)
        fullmeta = self.vector_meta.iloc[ilocs].assign(score=scores)
        return rescore_candidates(fullmeta, topk, **kwargs)

    def _rescore_avg_on_device(self, candidate_df, topk, aug_larger, vector2=None, aug_weight="level_max"):
        """rescore_candidates for the averaging aggregation (multiscale_index.py:379-403): frames in ascending
        dbidx order, per frame the first tile with the highest aggregated score, then the topk frames by
        np.argsort(-score) -- the per-frame work runs in one kernel launch instead of a pandas loop"""
        positions = np.sort(np.asarray(candidate_df.attrs["positions"], dtype=np.int64))
        minus = None
        if vector2 is not None:
            minus = self._dev.score_rows(vector2, self._candidate_rows(positions))
        assert aug_weight in ("level_max", "cont_weighted"), aug_weight  # score_frame2's `assert False`
        scores, rows = self._dev.rescore_avg(positions, aug_larger, minus, aug_weight=aug_weight)
        top = np.argsort(-scores.astype(np.float64))[:topk]
        return {"dbidxs": self._dbidx[positions[top]].astype("int"),
                "activations": ActivationFrames(self._box[rows[top]], self._row_dbidx[rows[top]], scores[top])}

    def rescore_avg_from_device_scores(self, candidate_df, topk, aug_larger, dev_scores_ptr: int):
        """rescore_candidates(fullmeta with score = the caller's float64 per-row scores, agg 'avg_score') for scores that
        are on the device already -- the graph loops' second stage (graph_based.py:100-108) in one kernel launch"""
        positions = np.sort(np.asarray(candidate_df.attrs["positions"], dtype=np.int64))
        scores, rows = self._dev.rescore_avg_f64(dev_scores_ptr, positions, aug_larger)
        top = np.argsort(-scores, kind="stable")[:topk]
        return {"dbidxs": self._dbidx[positions[top]].astype("int"),
                "activations": ActivationFrames(self._box[rows[top]], self._row_dbidx[rows[top]], scores[top])}

    def new_query(self):
        return BoxFeedbackQuery(self)

    def get_data(self, dbidx) -> pd.DataFrame:
        vmeta = self.vector_meta[self.vector_meta.dbidx == dbidx]
        return vmeta.assign(vectors=list(self.vectors[vmeta.index]))

    def subset(self, indices: BitMap) -> AccessMethod:
        mask = np.isin(self._row_dbidx, np.asarray(indices, dtype=np.int64))
        if mask.all():
            return self
        return MultiscaleIndex(embedding=self.embedding, vectors=self.vectors[mask],
                               vector_meta=self.vector_meta[mask].reset_index(drop=True), vec_index=None,
                               device=self.device)


class BoxFeedbackQuery(InteractiveQuery):
    index: MultiscaleIndex

    def __init__(self, db):
        super().__init__(db)
        assert self.index is not None
        self.all_dbidx = FrozenBitMap(self.index._dbidx)

    def query_random(self, batch_size):
        remaining = np.asarray(self.all_dbidx - self.returned, dtype=np.int64)
        return {"dbidxs": np.random.permutation(remaining)[:batch_size].astype("int"), "activations": None}

    def _matched_fast(self, target_description=None) -> pd.DataFrame:
        """match_labels_to_vectors without the per-image pandas joins: for every seen image the
        max IoU of each of its tiles with the image's accepted boxes; results are cached per
        (image, label content) because earlier rounds' labels do not change."""
        rows, miou = self._matched_arrays(target_description)
        idx = self.index
        if rows.shape[0] == 0:
            return pd.DataFrame({"dbidx": np.zeros(0, np.int64), "ys": np.zeros(0), "max_iou": np.zeros(0)},
                                index=pd.Index(np.zeros(0, dtype=np.int64)))
        return pd.DataFrame({"dbidx": idx._row_dbidx[rows], "ys": (miou > 0).astype("float"), "max_iou": miou},
                            index=pd.Index(rows))

    def _match_one(self, dbidx: int, target_description):
        """(tile rows of image dbidx, their max IoU with the image's accepted boxes), or None for an image the index lacks"""
        idx = self.index
        boxes = self.label_db.ldata[int(dbidx)] or []
        if target_description is not None:
            sel = [b for b in boxes if b.description == target_description]
        else:
            sel = [b for b in boxes if b.marked_accepted]
        pos = int(np.searchsorted(idx._dbidx, dbidx))
        if pos >= idx._dbidx.shape[0] or idx._dbidx[pos] != dbidx:
            return None
        rows = np.arange(idx._row_start[pos], idx._row_start[pos + 1])
        miou = np.zeros(rows.shape[0])
        if sel:
            t = idx._box[rows].astype(np.float64)
            g = np.array([[b.x1, b.y1, b.x2, b.y2] for b in sel], dtype=np.float32).astype(np.float64)
            w = np.clip(np.minimum(t[:, None, 2], g[None, :, 2]) - np.maximum(t[:, None, 0], g[None, :, 0]), 0, None)
            h = np.clip(np.minimum(t[:, None, 3], g[None, :, 3]) - np.maximum(t[:, None, 1], g[None, :, 1]), 0, None)
            inter = w * h
            union = ((t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1]))[:, None] + \
                    ((g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1]))[None, :] - inter
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = np.where(inter > 0, inter / union, 0.0)
            miou = iou.max(axis=1)
        return rows, miou

    def _matched_arrays(self, target_description=None):
        """(tile rows of every seen image, their max IoU with the image's accepted boxes) as plain READ-ONLY arrays, images in
        ascending dbidx order (label_db.get_seen()).  The label store logs which images' labels changed (LabelDB.changes): a
        call matches those and nothing else -- one image a round in a benchmark session, not every image seen so far -- and a
        call with nothing new returns the arrays of the call before."""
        import bisect
        ldb = self.label_db
        changes = getattr(ldb, "changes", None)
        if changes is None or len(ldb.ldata) != len(getattr(ldb, "stamp", ())):
            return self._matched_arrays_full(target_description)  # a label store without the change log
        states = self.__dict__.setdefault("_match_state", {})
        st = states.get(target_description)
        if st is None:
            st = states[target_description] = {"pos": 0, "entries": {}, "order": [], "cat": None}
        if st["cat"] is not None and st["pos"] == len(changes):
            return st["cat"]
        entries, order = st["entries"], st["order"]
        for dbidx in dict.fromkeys(changes[st["pos"]:]):  # (an image changed twice since is matched once)
            hit = self._match_one(dbidx, target_description)
            known = dbidx in entries
            if hit is None:
                if known:
                    del entries[dbidx]
                    order.remove(dbidx)
                continue
            entries[dbidx] = hit
            if not known:
                bisect.insort(order, dbidx)
        st["pos"] = len(changes)
        if order:
            rows = np.concatenate([entries[d][0] for d in order])
            miou = np.concatenate([entries[d][1] for d in order])
        else:
            rows, miou = np.zeros(0, dtype=np.int64), np.zeros(0)
        rows.setflags(write=False)
        miou.setflags(write=False)
        st["cat"] = (rows, miou)
        return st["cat"]

    def _matched_arrays_full(self, target_description=None):
        """_matched_arrays over every seen image (label stores without LabelDB.changes)"""
        rows_all, iou_all = [], []
        for dbidx in sorted(self.label_db.ldata):  # == label_db.get_seen() (ascending dbidx)
            hit = self._match_one(dbidx, target_description)
            if hit is not None:
                rows_all.append(hit[0])
                iou_all.append(hit[1])
        if not rows_all:
            return np.zeros(0, dtype=np.int64), np.zeros(0)
        return np.concatenate(rows_all), np.concatenate(iou_all)

    def getXy(self, get_positions=False, target_description=None):
        if get_positions:  # the graph loops' form: vector positions only, no frame to build
            rows, miou = self._matched_arrays(target_description)
            return rows[miou > 0], rows[~(miou > 0)]
        return self._matched_fast(target_description=target_description)  # columns dbidx, ys, max_iou
