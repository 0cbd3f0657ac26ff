"""Result-quality metrics over the positions of the hits (seesaw/metrics.py:8-137)."""
import math

import numpy as np


def average_precision(hit_indices, *, npositive, max_results=None, average_reciprocal_gap=False):
    """mean over the first `max_results` positives of (#found so far) / rank; positives never
    reached contribute 0."""
    assert npositive > 0
    max_results = npositive if max_results is None else min(npositive, max_results)
    hits = np.asarray(hit_indices)[:max_results]
    ranks = hits + 1
    denom = np.full(max_results, np.inf)
    if average_reciprocal_gap:
        denom[: hits.shape[0]] = np.diff(np.concatenate(([0], ranks)))
        num = 1.0
    else:
        denom[: hits.shape[0]] = ranks
        num = np.arange(max_results) + 1
    return np.mean(num / denom)


def average_reciprocal_gap(*args, **kwargs):
    return average_precision(*args, **kwargs, average_reciprocal_gap=True)


def dcg_score(hit_indices):
    return (1.0 / np.log2(np.asarray(hit_indices) + 2)).sum()


def rank_of_kth(hit_indices, *, ntotal, k):
    if k > ntotal:
        return None
    return math.inf if hit_indices.shape[0] < k else hit_indices[k - 1] + 1


def rank_kth(hit_indices, *, ntotal, ks):
    ans = np.ones_like(ks, dtype=float)
    found = ks <= hit_indices.shape[0]
    ans[~found] = np.inf
    ans[ks > ntotal] = np.nan
    ans[found] = hit_indices[ks[found] - 1] + 1
    return ans


def best_possible_hits(nseen, npositive):
    return np.arange(min(nseen, npositive))


def ndcg_score(hit_indices, *, nseen, npositive):
    return dcg_score(hit_indices) / dcg_score(best_possible_hits(nseen, npositive))


def normalizedAP(hit_indices, *, nseen, npositive, max_results=None):
    best = average_precision(best_possible_hits(nseen, npositive), npositive=npositive, max_results=max_results)
    return average_precision(hit_indices, npositive=npositive, max_results=max_results) / best


def compute_metrics(*, hit_indices, batch_size, nseen, ntotal, max_results):
    ap = average_precision(hit_indices, npositive=ntotal, max_results=max_results)
    ndcg = ndcg_score(hit_indices, nseen=nseen, npositive=ntotal)
    r1, r2, r3, r10 = rank_kth(hit_indices, ntotal=ntotal, ks=np.array([1, 2, 3, 10]))
    return dict(nfound=hit_indices.shape[0], ndcg_score=ndcg, average_precision=ap, rank_first=r1,
                reciprocal_rank=1.0 / r1, rank_second=r2, rank_third=r3, rank_tenth=r10)
