"""Efficient non-myopic search (seesaw/research/active_search/efficient_nonmyopic_search.py): which node to show next
so that the expected number of positives over the next `reward_horizon` picks is largest, planning one or two
steps exactly and the rest greedily.

implementation='vectorized' is the reference's production form (_opt_expected_utility_helper_lknn2 + _top_sum,
:94-205); its N x (K + 2D) argsort is ssw_lknn_top_sum here.  implementation='loop' is the recursive definition
(:19-90), kept for small graphs: it is what the vectorised form is checked against."""
from __future__ import annotations

import math

import numpy as np

from ...bitmap import BitMap
from .common import ProbabilityModel, Result


def _expected_utility_approx(t: int, model: ProbabilityModel):
    assert t > 0
    idxs, scores = model.top_k_remaining(top_k=t)
    return Result(value=scores.sum(), index=idxs[0], pruned_fraction=None)


def _opt_expected_utility_helper(*, i: int, lookahead_limit: int, t: int, model: ProbabilityModel, pruning_on: bool):
    """expected utility at horizon t with an exact look-ahead of `lookahead_limit` steps (recursive form)"""
    assert 0 <= i < lookahead_limit
    if i == lookahead_limit - 1:
        return _expected_utility_approx(t - i, model)
    idxs = model.dataset.remaining_indices()
    p1 = model.predict_proba(idxs).reshape(-1, 1)
    probs = np.concatenate([1 - p1, p1], axis=-1)

    def solve(idx):
        u0 = _opt_expected_utility_helper(i=i + 1, lookahead_limit=lookahead_limit, t=t, model=model.condition(idx, 0),
                                          pruning_on=pruning_on)
        u1 = _opt_expected_utility_helper(i=i + 1, lookahead_limit=lookahead_limit, t=t, model=model.condition(idx, 1),
                                          pruning_on=pruning_on)
        return np.array([u0.value, u1.value])

    pruned_fraction = 0.0
    if pruning_on:
        pbound = model.probability_bound(1)
        top_idxs, top_ps = model.top_k_remaining(top_k=(t - i))
        assert top_ps.shape[0] == t - i
        upper = p1 * (1 + (t - i) * pbound) + (1 - p1) * top_ps.sum()
        lower = solve(top_idxs[0]) @ np.array([1 - top_ps[0], top_ps[0]])
        pruned = (upper < lower).squeeze()
        pruned_fraction = pruned.sum() / pruned.shape[0]
        gone = BitMap(int(idxs[int(pos)]) for pos in np.where(pruned)[0])
        idxs = idxs - gone
        probs = probs[~pruned]
    values = np.zeros_like(probs)
    for j, idx in enumerate(idxs):
        values[j, :] = solve(idx)
    expected = (probs * (values + np.array([0, 1]).reshape(1, -1))).sum(axis=-1)
    pos = np.argmax(expected)
    return Result(value=expected[pos], index=idxs[int(pos)], pruned_fraction=pruned_fraction)


def _opt_expected_utility_helper_lknn2(*, i: int, lookahead_limit: int, t: int, model, pruning_on: bool):
    assert i == 0 and lookahead_limit <= 2 and t >= lookahead_limit
    assert ((0 < model.gamma) & (model.gamma < 1)).all()
    assert (model.numerators <= model.denominators).all()
    if lookahead_limit == 2:
        best, value = model.top_sum(K=t - 1)
        return Result(value=value, index=best, pruned_fraction=0.0)
    numer = model.numerators + model.gamma
    numer[np.asarray(model.dataset.seen_indices, dtype=np.int64)] = -math.inf
    scores = numer / (model.denominators + 1)
    best = np.nanargmax(scores)
    return Result(value=scores[best], index=best, pruned_fraction=0.0)


def efficient_nonmyopic_search(model: ProbabilityModel, *, reward_horizon: int, lookahead_limit: int, pruning_on: bool,
                               implementation: str) -> Result:
    assert reward_horizon > 0
    assert 1 <= lookahead_limit <= 2, "implementation assumes at most 1 lookahead (pruning)"
    assert lookahead_limit <= reward_horizon
    if implementation == "vectorized":
        return _opt_expected_utility_helper_lknn2(i=0, lookahead_limit=lookahead_limit, t=reward_horizon, model=model,
                                                  pruning_on=pruning_on)
    assert implementation == "loop", implementation
    return _opt_expected_utility_helper(i=0, lookahead_limit=lookahead_limit, t=reward_horizon, model=model,
                                        pruning_on=pruning_on)
