"""CPU ORACLE -- test infrastructure, never the product path.

numpy / scipy / torch-CPU restatements of the reference's hot path, each citing the
reference file:line it follows (paths relative to the reference root, orm011/seesaw).
Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may
import this module; nothing under ``seesaw_amd/`` does.

Parity status: every function here is pinned against outputs of the reference itself
(``tests/golden/*.npz``, produced in the build container by ``oracle/gen_golden.py``,
which imports the reference from /root/reference) and against the reference's own
known-answer tests where it has them (rank-loss table, distinct_topk_positions).
CLIP is the exception: the reference holds no fixture for it and delegates the
arithmetic to ``transformers`` (parity unpinned by the reference): its oracle is the
in-container ``transformers.CLIPModel`` (f32, torch-CPU, seeded random-init weights),
built and compared in ``tests/test_clip_gpu.py`` (fixture ``models``); there is no
restatement of CLIP in this directory.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read()
    except OSError:
        return False


def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/ssw_oracle.c with gcc (rebuilds when the host's FMA support differs
    from the machine the library was last built on)."""
    out = os.path.join(_HERE, "_build", "libssw_oracle.so")
    stamp = os.path.join(_HERE, "_build", f".fma_{int(_cpu_has_fma())}")
    cmd = ["make", "-C", _HERE]
    rebuild = force or not os.path.exists(stamp)
    if rebuild:
        cmd.append("-B")
    subprocess.run(cmd, check=True, capture_output=True)
    if rebuild:  # (the stamps are touched only then: two ranks of a gloo test come through here at the same time)
        for f in os.listdir(os.path.join(_HERE, "_build")):
            if f.startswith(".fma_") and os.path.join(_HERE, "_build", f) != stamp:
                try:
                    os.remove(os.path.join(_HERE, "_build", f))
                except FileNotFoundError:
                    pass
        open(stamp, "w").close()
    return out


def c_lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_c_oracle())
        f32p = ctypes.POINTER(ctypes.c_float)
        lib.ssw_oracle_scores_kernel_order.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int32, f32p]
        lib.ssw_oracle_scores_kernel_order.restype = None
        lib.ssw_oracle_scores_f64.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.POINTER(ctypes.c_double)]
        lib.ssw_oracle_scores_f64.restype = None
        lib.ssw_oracle_synth_rows.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int32, f32p]
        lib.ssw_oracle_synth_rows.restype = None
        lib.ssw_oracle_topk_images.argtypes = [
            f32p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.c_int64,
            ctypes.POINTER(ctypes.c_uint8), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), f32p,
            ctypes.POINTER(ctypes.c_int64)]
        lib.ssw_oracle_topk_images.restype = ctypes.c_int64
        _LIB = lib
    return _LIB


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# --------------------------------------------------------------------------------------
# scan
# --------------------------------------------------------------------------------------
def scores_reference(vectors: np.ndarray, vector: np.ndarray) -> np.ndarray:
    """`scores = vectors @ vector.reshape(-1)` -- multiscale_index.py:171 / :285,
    coarse_index.py:38 (BLAS sgemv; summation order unspecified)."""
    return vectors @ vector.reshape(-1)


def scores_kernel_order(vectors: np.ndarray, vector: np.ndarray) -> np.ndarray:
    """Same dot products in the fixed order of seesaw_amd/csrc/scan.hip (bit-exact twin)."""
    X = np.ascontiguousarray(vectors, dtype=np.float32)
    q = np.ascontiguousarray(vector.reshape(-1), dtype=np.float32)
    assert X.shape[1] == q.shape[0] and X.shape[1] % 256 == 0
    out = np.empty(X.shape[0], dtype=np.float32)
    c_lib().ssw_oracle_scores_kernel_order(_f32p(X), _f32p(q), X.shape[0], X.shape[1], _f32p(out))
    return out


def scores_f64(vectors: np.ndarray, vector: np.ndarray) -> np.ndarray:
    X = np.ascontiguousarray(vectors, dtype=np.float32)
    q = np.ascontiguousarray(vector.reshape(-1), dtype=np.float32)
    out = np.empty(X.shape[0], dtype=np.float64)
    c_lib().ssw_oracle_scores_f64(_f32p(X), _f32p(q), X.shape[0], X.shape[1],
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out


def get_top_exact(vector: np.ndarray, *, vectors: np.ndarray):
    """_get_top_exact -- multiscale_index.py:170-175: full descending argsort."""
    scores = vectors @ vector.reshape(-1)
    vec_idxs = np.argsort(-scores)
    vec_scores = scores[vec_idxs]
    return vec_idxs, vec_scores


def distinct_topk_positions(dbidxs: np.ndarray, topk: int) -> np.ndarray:
    """distinct_topk_positions -- multiscale_index.py:177-180."""
    _, index = np.unique(dbidxs, return_index=True)
    return np.sort(index)[:topk]


def get_top_dbidxs(*, vec_idxs, scores, row_dbidx: np.ndarray, exclude, topk: int):
    """_get_top_dbidxs -- multiscale_index.py:189-199 without the pandas wrapper:
    map sorted rows to dbidx, drop excluded, keep first occurrence per image, cut at topk.
    Returns (dbidx[<=topk], max_score[<=topk], row position of that max)."""
    dbidx = row_dbidx[vec_idxs]
    excl = np.fromiter(exclude, dtype=np.int64) if exclude is not None else np.zeros(0, np.int64)
    mask = ~np.isin(dbidx, excl)
    new_dbidx = dbidx[mask]
    new_scores = scores[mask]
    new_rows = np.asarray(vec_idxs)[mask]
    pos = distinct_topk_positions(new_dbidx, topk=topk)
    return new_dbidx[pos], new_scores[pos], new_rows[pos]


def topk_images_reference(vectors, vector, row_dbidx, exclude, topk):
    """_query_prelim(force_exact=True) -- multiscale_index.py:291-312."""
    vec_idxs, vec_scores = get_top_exact(vector, vectors=vectors)
    return get_top_dbidxs(vec_idxs=vec_idxs, scores=vec_scores, row_dbidx=row_dbidx,
                          exclude=exclude, topk=topk)


def topk_images_tiebreak(scores: np.ndarray, row2image, n_images: int, excluded_positions, k: int):
    """The same selection with the HIP path's deterministic tie rule (score desc, image
    position asc, row asc) applied to a GIVEN score vector -- C restatement."""
    s = np.ascontiguousarray(scores, dtype=np.float32)
    n = s.shape[0]
    r2i = None
    if row2image is not None:
        r2i_arr = np.ascontiguousarray(row2image, dtype=np.int32)
        r2i = r2i_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    mask = np.zeros(max(n_images, 1), dtype=np.uint8)
    ex = np.asarray(list(excluded_positions) if excluded_positions is not None else [], dtype=np.int64)
    if ex.size:
        mask[ex] = 1
    out_i = np.empty(k, dtype=np.int64)
    out_s = np.empty(k, dtype=np.float32)
    out_r = np.empty(k, dtype=np.int64)
    cnt = c_lib().ssw_oracle_topk_images(
        _f32p(s), n, r2i, n_images, mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), k,
        out_i.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _f32p(out_s),
        out_r.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return out_i[:cnt], out_s[:cnt], out_r[:cnt]


def rounding_band(vectors: np.ndarray, vector: np.ndarray) -> float:
    """Largest |f32 dot - exact dot| any summation order of `dim` products can reach
    (standard gamma_n bound), used to decide which top-k boundaries are ambiguous
    between two valid f32 orders (BLAS vs HIP kernel)."""
    dim = vectors.shape[1]
    eps = np.finfo(np.float32).eps
    bound = float(dim) * eps * float(np.max(np.abs(vectors) @ np.abs(vector.reshape(-1))))
    return 2.0 * bound


def check_topk_against_reference(got_images, ref_scores_f32: np.ndarray, row2image, excluded_positions,
                                 k: int, band: float):
    """Set-parity rule (BASELINE.md section 3 / SURVEY section 7 'Hard parts'): the image
    SET returned must equal the reference's set whenever the gap between the k-th and
    (k+1)-th best image exceeds the rounding band; otherwise the symmetric difference
    must lie entirely within the band around the k-th score.  Returns (ok, message)."""
    n = ref_scores_f32.shape[0]
    r2i = np.arange(n) if row2image is None else np.asarray(row2image)
    n_img = int(r2i.max()) + 1 if n else 0
    best = np.full(n_img, -np.inf, dtype=np.float64)
    np.maximum.at(best, r2i, ref_scores_f32.astype(np.float64))
    if excluded_positions is not None and len(excluded_positions):
        best[np.asarray(list(excluded_positions), dtype=np.int64)] = -np.inf
    order = np.argsort(-best, kind="stable")
    valid = int(np.isfinite(best).sum())
    kk = min(k, valid)
    ref_set = set(order[:kk].tolist())
    got_set = set(int(x) for x in got_images)
    if len(got_set) != kk:
        return False, f"expected {kk} images, got {len(got_set)}"
    if ref_set == got_set:
        return True, "exact"
    kth = best[order[kk - 1]]
    diff = ref_set ^ got_set
    worst = max(abs(best[i] - kth) for i in diff)
    if worst <= band:
        return True, f"ambiguous boundary: {len(diff)} images within band {band:.3e}"
    return False, f"set differs outside the rounding band: worst {worst:.3e} > {band:.3e}"


# --------------------------------------------------------------------------------------
# synthetic data (the benchmark's input generator; not part of the reference)
# --------------------------------------------------------------------------------------
def synth_rows(seed: int, first_row: int, n: int, dim: int = 512) -> np.ndarray:
    """Bit-identical CPU twin of seesaw_amd/csrc/rng.hip."""
    out = np.empty((n, dim), dtype=np.float32)
    c_lib().ssw_oracle_synth_rows(ctypes.c_uint64(seed), first_row, n, dim, _f32p(out))
    return out


def synth_query(seed: int, dim: int = 512) -> np.ndarray:
    """Unit-norm query drawn from the same generator (row `2**40 + seed` of stream 0xC0FFEE)."""
    return synth_rows(0xC0FFEE, (1 << 40) + seed, 1, dim)[0]


# --------------------------------------------------------------------------------------
# label propagation
# --------------------------------------------------------------------------------------
def exact_knn(vectors: np.ndarray, k: int):
    """compute_exact_knn (seesaw/knn_graph.py:170-191) restated row by row: every row is scanned
    against all rows (kernel-order f32 scores), the k+1 best rows INCLUDING itself are kept in
    (score desc, row id asc) order.  The reference orders by `1 - X @ X.T` with BLAS sums and an
    unstable argsort, so it agrees with this up to f32 rounding of near-ties (checked in
    tests/test_oracle_cpu.py against its golden output).  Returns (dst int32 [n,k+1], score f32)."""
    X = np.ascontiguousarray(vectors, dtype=np.float32)
    n = X.shape[0]
    k1 = min(k + 1, n)
    dst = np.empty((n, k1), dtype=np.int32)
    score = np.empty((n, k1), dtype=np.float32)
    ids = np.arange(n)
    for i in range(n):
        s = scores_kernel_order(X, X[i])
        order = np.lexsort((ids, -s.astype(np.float64)))[:k1]
        dst[i], score[i] = order, s[order]
    return dst, score


def label_propagation(W, *, label_ids, label_values, reg_lambda, reg_values=None, start_value=None,
                      max_iter=300, epsilon=1e-5):
    """LabelPropagation.fit_transform -- seesaw/label_propagation.py:45-79 with _step (:30-43):
    weighted = W @ f + lambda*reg; f' = weighted / (W.sum(0) + lambda); f'[ids] = values;
    stop when max((f'-f)^2) < eps returning f (not f'); otherwise f <- f'.
    Returns (scores, sweeps_run, converged)."""
    n = W.shape[0]
    weight_sum = np.asarray(W.sum(0)).reshape(-1)
    if reg_values is None:
        assert reg_lambda == 0
        reg = np.zeros(n)
    else:
        reg = reg_values
    if start_value is not None:
        old = np.array(start_value, dtype=np.float64)
    elif reg_values is not None:
        old = np.array(reg_values, dtype=np.float64)
    else:
        old = np.zeros(n)
    label_ids = np.asarray(label_ids, dtype=np.int64).reshape(-1)
    old[label_ids] = label_values
    low = min(0, reg.min())
    high = max(1.0, reg.max())
    sweeps, converged = 0, False
    for _ in range(max_iter):
        new = (W @ old + reg_lambda * reg) / (weight_sum + reg_lambda)
        assert (new >= low).all() and (new <= high).all()
        new[label_ids] = label_values
        sweeps += 1
        if np.max((new - old) ** 2) < epsilon:
            converged = True
            break
        old = new
    return old, sweeps, converged


# ---- avg_score aggregation: score_frame2 / box_join ---------------------------------------
def box_iou_f32(boxes: np.ndarray):
    """all-pairs IoU of one image's tile boxes [T, 4] = x1, y1, x2, y2 in f32, op for op as box_utils.box_iou
    (box_utils.py:336-350) forms it through torchvision's _box_inter_union on float32 tensors:
    area = (x2-x1)*(y2-y1); wh = clamp(min(rb) - max(lt), 0); inter = w*h; union = (a_i + a_j) - inter."""
    b = np.asarray(boxes, dtype=np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    w = np.maximum(np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0]), np.float32(0))
    h = np.maximum(np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1]), np.float32(0))
    inter = (w * h).astype(np.float32)
    union = ((area[:, None] + area[None, :]).astype(np.float32) - inter).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(np.float32)


def _kahan_mean_f32(values):
    """pandas' groupby mean of a float32 column: Kahan-compensated f32 sum in row order, f32 division"""
    s = np.float32(0)
    c = np.float32(0)
    for x in values:
        y = np.float32(np.float32(x) - c)
        t = np.float32(s + y)
        c = np.float32(np.float32(t - s) - y)
        s = t
    return np.float32(s / np.float32(len(values)))


def avg_score_image(boxes: np.ndarray, zoom: np.ndarray, scores: np.ndarray, aug_larger: str):
    """score_frame2 with aug_weight='level_max' for ONE image (multiscale_index.py:112-150): tile i's score becomes
    the mean, over the zoom levels z allowed by aug_larger, of the score of the tile of level z overlapping i most
    (first such tile on ties; only pairs with IoU > 0 take part).  -> (index of the first tile with the highest
    aggregated score, that score f32, all aggregated scores)."""
    iou = box_iou_f32(boxes)
    zoom = np.asarray(zoom).astype(np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    T = zoom.shape[0]
    agg = np.empty(T, dtype=np.float32)
    for i in range(T):
        winners = []
        for z in np.unique(zoom):  # ascending: the order pandas sums the groups' rows in
            if aug_larger == "greater" and z < zoom[i]:
                continue
            if aug_larger == "adjacent" and z != zoom[i]:
                continue
            assert aug_larger in ("all", "greater", "adjacent")
            js = np.nonzero((zoom == z) & (iou[i] > 0))[0]
            if js.size:
                winners.append(scores[js[np.argmax(iou[i, js])]])  # argmax: first maximum
        agg[i] = _kahan_mean_f32(winners) if winners else np.float32(np.nan)
    best = int(np.flatnonzero(agg == np.nanmax(agg))[0])
    return best, agg[best], agg


def rescore_avg_score(row_dbidx, boxes, zoom, scores, topk: int, aug_larger: str):
    """rescore_candidates for agg_method='avg_score' (multiscale_index.py:379-403): candidate tiles (sorted by
    image) -> (dbidxs [topk], best tile position in the inputs [topk], aggregated score [topk])."""
    row_dbidx = np.asarray(row_dbidx)
    ids, starts = np.unique(row_dbidx, return_index=True)
    bounds = list(starts) + [row_dbidx.shape[0]]
    best_rows, best_scores = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        j, sc, _ = avg_score_image(boxes[a:b], zoom[a:b], scores[a:b], aug_larger)
        best_rows.append(a + j)
        best_scores.append(sc)
    dbscores = np.asarray(best_scores, dtype=np.float64)
    top = np.argsort(-dbscores)[:topk]
    return ids[top], np.asarray(best_rows)[top], np.asarray(best_scores, dtype=np.float32)[top]


# ---- seeded labelled sets shared by the golden generator and the tests (inputs regenerate from three integers)
def labelled_set(seed, n, n_pos, dim=512, q_noise=0.8):
    """A small labelled set shaped like q.getXy(): tile vectors of seen images."""
    rng = np.random.default_rng(seed)
    target = synth_query(seed)
    X = synth_rows(seed, 0, n, dim)
    y = np.zeros(n)
    pos = rng.choice(n, n_pos, replace=False)
    X[pos] = X[pos] + 0.6 * target
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    y[pos] = 1.0
    q = target + q_noise * synth_query(seed + 77)
    q = (q / np.linalg.norm(q)).astype(np.float32)
    return X, y, q


# ---- L-KNN two-step look-ahead: _top_sum ------------------------------------------------------
def lknn_top_sum(numerators, denominators, neighbor_ids_sorted, K):
    """_top_sum (seesaw/research/active_search/efficient_nonmyopic_search.py:94-169) row by row.  numerators already
    hold + gamma (-inf at labelled nodes), denominators + 1.  For node i the candidates are the K + D globally best
    scores -- minus i itself and minus those of its neighbours that are among them -- plus the neighbours' scores
    had i been labelled y (i itself excluded); E_y = sum of the K best (numpy's row sum of the descending values);
    value = s (1 + E_1) + (1 - s) E_0."""
    num, den = np.asarray(numerators, np.float64), np.asarray(denominators, np.float64)
    nbr = np.asarray(neighbor_ids_sorted)
    N, D = nbr.shape
    scores = num / den
    top = np.argsort(scores)[-(K + D):]
    top_scores = scores[top]
    new_den = den + 1
    given = {0: num / new_den, 1: (num + 1) / new_den}
    out = np.empty(N)
    for i in range(N):
        keep = ~(np.isin(top, nbr[i]) | (top == i))
        e = {}
        for y in (0, 1):
            ns = given[y][nbr[i]].copy()
            ns[nbr[i] == i] = -np.inf
            cand = np.concatenate([np.where(keep, top_scores, -np.inf), ns])
            best = np.sort(cand)[::-1][:K].reshape(1, -1).copy()
            e[y] = best.sum(axis=1)[0]
        with np.errstate(invalid="ignore"):
            out[i] = scores[i] * (1 + e[1]) + (1 - scores[i]) * e[0]
    return out


# ---- seeded images of the tiling fixture (tests/golden/tiling.npz; oracle/gen_golden.py gen_tiling) ----------------------
TILING_SIZES = [(640, 480), (224, 224), (500, 375), (1000, 300), (225, 900), (100, 80), (1344, 896)]  # (width, height)


def tiling_image(w, h, seed):
    """a seeded RGB image with structure at several scales (a resize of pure noise would hide an off-by-one in a box)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx // 16 + yy // 16) % 2) * 255], axis=2)
    noise = rng.integers(-40, 41, size=(h, w, 3))
    return np.clip(base + noise, 0, 255).astype(np.uint8)
