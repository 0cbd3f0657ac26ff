"""CPU ORACLE (test infrastructure / bench.py cpu_baseline leg): one feedback session with
the reference's own numeric expressions on the host cores -- numpy `X @ q` + full
`np.argsort`, scipy CSR label propagation, torch-CPU L-BFGS -- driven like
seesaw_bench.benchmark_loop (seesaw/seesaw_bench.py:278-355): next -> simulated labels ->
refine, timing every iteration.  Shares no code with seesaw_amd's numeric path."""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp

from . import feedback_oracle as fo
from . import seesaw_oracle as orc


def _rbf_weight_matrix(knn_df, edist):
    """get_weight_matrix(symmetric=True) restated with scipy (knn_graph.py:31-104)."""
    src, dst = knn_df.src_vertex.values.astype(np.int64), knn_df.dst_vertex.values.astype(np.int64)
    n = int(src.max()) + 1
    w = np.exp(-(knn_df.distance.values.astype(np.float64) / edist))
    rows, cols = np.concatenate([src, dst]), np.concatenate([dst, src])
    ws = sp.coo_array((np.concatenate([w, w]), (rows, cols)), shape=(n, n)).tocsr()
    ct = sp.coo_array((np.ones(2 * w.shape[0]), (rows, cols)), shape=(n, n)).tocsr()
    ws.sum_duplicates(); ct.sum_duplicates(); ws.sort_indices(); ct.sort_indices()
    out = sp.csr_array((ws.data / ct.data, ws.indices, ws.indptr), shape=(n, n))
    out.setdiag(0.0)
    out.sort_indices()
    return out


def run_session(vectors, vector_meta, box_data, category, qvec, *, loop="multi_reg", n_batches=30, shortlist=50,
                max_results=None, knn_df=None, edist=0.05, reg_norm_lambda=100.0, max_iter=200, sample_size=10000):
    """Returns dict(nfound, nseen, latencies, shown).  loop in {plain, multi_reg, knn_prop2, pseudo_lr}.
    pseudo_lr (loops/pseudo_lr.py:10-54): label propagation, then an (unregularised: LoopState.tvec is never set)
    logistic fit on the real labels plus `sample_size` pseudo-labelled vectors drawn with np.random.permutation;
    the graph ranks the batches until both classes have a real label (switch_over)."""
    row_dbidx = vector_meta.dbidx.values.astype(np.int64)
    boxes = box_data[box_data.category == category]
    positives = set(boxes.dbidx.tolist())
    max_results = len(positives) if max_results is None else min(len(positives), max_results)
    meta_xyxy = vector_meta[["x1", "y1", "x2", "y2"]].values.astype(np.float64) if "x1" in vector_meta else None
    returned, shown, latencies = [], [], []
    labelled_rows, labelled_y = [], []
    curr = qvec.reshape(-1).astype(np.float32)
    q0 = curr.copy()
    lp_scores = None
    W = None
    graph = loop in ("knn_prop2", "pseudo_lr")
    if graph:
        W = _rbf_weight_matrix(knn_df, edist)
        from scipy.special import expit
        prior = expit(10.0 * ((vectors @ q0).astype(np.float64) - 0.4))
        lp_scores = prior
        is_labeled = np.zeros(vectors.shape[0], dtype=bool)
        labels = np.zeros(vectors.shape[0])
    nfound = 0
    started = False
    for it in range(1, n_batches + 1):
        t0 = time.time()
        # ---- next(): scan + full sort + distinct non-returned images, best tile per image
        both = bool(labelled_y) and (np.asarray(labelled_y) > 0).any() and (np.asarray(labelled_y) == 0).any()
        if (loop == "knn_prop2" and started) or (loop == "pseudo_lr" and started and not both):
            s = np.where(is_labeled, -np.inf, lp_scores)
            order = np.argsort(-s)
            d, sc, rows = orc.get_top_dbidxs(vec_idxs=order, scores=s[order], row_dbidx=row_dbidx, exclude=returned,
                                             topk=shortlist)
        else:
            d, sc, rows = orc.topk_images_reference(vectors, curr, row_dbidx, returned, shortlist)
        if d.shape[0] == 0:
            break
        img = int(d[0])  # batch_size 1: the best image of the shortlist (plain_score aggregation)
        returned.append(img)
        shown.append(img)
        # ---- simulated user (fill_imdata, seesaw_bench.py:237-273: one np.random.rand per ground-truth box of the image,
        # the box_drop_prob draw -- drawn even at probability 0, so it moves numpy's global stream, which
        # pseudo_lr's np.random.permutation reads later)
        hit = img in positives
        if hit:
            np.random.rand(int((boxes.dbidx.values == img).sum()))
        nfound += int(hit)
        img_rows = np.nonzero(row_dbidx == img)[0]
        if hit and meta_xyxy is not None:
            gt = boxes[boxes.dbidx == img][["x1", "y1", "x2", "y2"]].values.astype(np.float64)
            tb = meta_xyxy[img_rows]
            iw = np.clip(np.minimum(tb[:, None, 2], gt[None, :, 2]) - np.maximum(tb[:, None, 0], gt[None, :, 0]), 0, None)
            ih = np.clip(np.minimum(tb[:, None, 3], gt[None, :, 3]) - np.maximum(tb[:, None, 1], gt[None, :, 1]), 0, None)
            ys = ((iw * ih).max(axis=1) > 0).astype(np.float64)
        else:
            ys = np.full(img_rows.shape[0], float(hit))
        labelled_rows.extend(img_rows.tolist())
        labelled_y.extend(ys.tolist())
        if nfound >= max_results or it == n_batches:
            break
        # ---- refine()
        started = True
        rows_a, y_a = np.asarray(labelled_rows), np.asarray(labelled_y)
        if loop == "multi_reg":
            X = vectors[rows_a]
            coeff, _ = fo.multireg_fit(X, y_a, row_dbidx[rows_a], q0, np.zeros((vectors.shape[1],) * 2, np.float32),
                                       loss_type="ce_loss", l_norm=reg_norm_lambda, l_data=0.0, l_query=0.0,
                                       max_iter=max_iter)
            curr = coeff
        elif graph:
            is_labeled[rows_a] = True
            labels[rows_a] = y_a
            if (y_a == 0).any():
                ids = np.nonzero(is_labeled)[0]
                lp_scores, _, _ = orc.label_propagation(W, label_ids=ids, label_values=labels[ids], reg_lambda=1.0,
                                                        reg_values=prior, start_value=prior)
            if loop == "pseudo_lr":  # makeXy (loops/util.py:4-23) + LogisticRegressionPT.fit (class_weights 1.0)
                import torch
                unl = np.nonzero(~is_labeled)[0]
                pick = unl[np.random.permutation(unl.shape[0])[:sample_size]]
                Xs = np.concatenate((vectors[is_labeled], vectors[pick]))
                ys = np.concatenate((labels[is_labeled], lp_scores[pick]))
                w0 = torch.nn.Linear(vectors.shape[1], 1, bias=False).weight.detach().numpy().reshape(-1)
                coeff, _ = fo.logreg_fit(Xs, ys, None, w0=w0, reg_lambda=1.0, class_weights=1.0,
                                         sample_weights=np.ones(ys.shape[0]), max_iter=max_iter, reg_kind=None)
                curr = coeff
        latencies.append(time.time() - t0)
    return dict(nfound=nfound, nseen=len(shown), latencies=latencies, shown=shown)
