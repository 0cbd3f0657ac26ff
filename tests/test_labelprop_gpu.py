"""GPU parity of label propagation (ssw_labelprop_run through seesaw_amd.label_propagation)
against outputs captured from the reference (tests/golden/labelprop.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _W(g, name="e05"):
    n = int(g["n"])
    return sp.csr_array((g[f"{name}_data"], g[f"{name}_indices"].astype(np.int32), g[f"{name}_indptr"]), shape=(n, n))


def test_fit_transform_bit_exact_vs_reference_golden():
    from seesaw_amd.label_propagation import LabelPropagation
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    W = _W(g)
    for r in range(int(g["n_runs"])):
        lam = float(g[f"run{r}_lam"])
        start = g[f"run{r}_start"]
        lp = LabelPropagation(W, reg_lambda=lam, max_iter=300)
        out = lp.fit_transform(label_ids=g[f"run{r}_ids"], label_values=g[f"run{r}_vals"],
                               reg_values=start if lam > 0 else None, start_value=start)
        assert lp.last_sweeps == int(g[f"run{r}_steps"]), (r, lp.last_sweeps)
        assert np.array_equal(out, g[f"run{r}_out"]), (r, np.abs(out - g[f"run{r}_out"]).max())
        lp.close()


def test_ranker_matches_reference_golden():
    from seesaw_amd.research.knn_methods import LabelPropagationRanker2
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    ranker = LabelPropagationRanker2(weight_matrix=_W(g), normalize_scores=False, sigmoid_before_propagate=True,
                                     calib_a=10.0, calib_b=-0.2, prior_weight=1.0)
    ranker.set_base_scores(g["rk_base_scores"])
    ids0, sc0 = ranker.top_k(k=20)
    assert np.array_equal(ids0, g["rk_top0_ids"]) and np.allclose(sc0, g["rk_top0_scores"], rtol=0, atol=1e-12)
    ranker.update(g["rk_upd_ids"], g["rk_upd_labels"])
    assert np.array_equal(ranker.current_scores(), g["rk_scores1"])
    ids1, sc1 = ranker.top_k(k=20)
    assert np.array_equal(ids1, g["rk_top1_ids"]) and np.array_equal(sc1, g["rk_top1_scores"])


@pytest.mark.parametrize("n,deg", [(1, 1), (257, 3), (5000, 20), (40000, 12)])
def test_random_graphs_vs_oracle(oracle, n, deg):
    from seesaw_amd.label_propagation import LabelPropagation
    rng = np.random.default_rng(n)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, n * deg)
    vals = rng.uniform(0.05, 1.0, n * deg)
    if n > 300:  # a hub row far longer than one LDS chunk, and an empty row
        hub = rng.integers(0, n, 9000)
        rows = np.concatenate([rows[rows != 7], np.full(9000, 5)])
        cols = np.concatenate([cols[: rows.shape[0] - 9000], hub])
        vals = np.concatenate([vals[: rows.shape[0] - 9000], rng.uniform(0.05, 1.0, 9000)])
    A = sp.coo_array((vals, (rows, cols)), shape=(n, n)).tocsr()
    # symmetric (column sums == row sums, so every sweep is a weighted average and the
    # reference's bound asserts hold) with a small diagonal so no normaliser is zero
    W = (A + A.T + sp.eye_array(n, format="csr") * 0.01).tocsr()
    W.sum_duplicates()
    W.sort_indices()
    prior = rng.uniform(0, 1, n)
    ids = rng.choice(n, size=min(n, 25), replace=False)
    lab = rng.integers(0, 2, ids.shape[0]).astype(np.float64)
    for lam, max_iter in [(0.0, 300), (2.0, 300), (1.0, 3), (1.0, 0)]:
        ref, sweeps, conv = oracle.label_propagation(W, label_ids=ids, label_values=lab, reg_lambda=lam,
                                                     reg_values=prior, start_value=prior, max_iter=max_iter)
        lp = LabelPropagation(W, reg_lambda=lam, max_iter=max_iter)
        out = lp.fit_transform(label_ids=ids, label_values=lab, reg_values=prior, start_value=prior)
        assert lp.last_sweeps == sweeps and lp.last_converged == conv
        assert np.array_equal(out, ref), np.abs(out - ref).max()
        lp.close()


def test_bound_violation_is_an_error():
    from seesaw_amd._lib import SeesawHipError
    from seesaw_amd.label_propagation import LabelPropagation
    W = sp.csr_array(np.array([[0.0, 1.0], [1.0, 0.0]]))
    lp = LabelPropagation(W, reg_lambda=0.0, max_iter=5)
    with pytest.raises(SeesawHipError):  # start value outside [0, 1]: the reference asserts
        lp.fit_transform(label_ids=np.zeros(0, np.int64), label_values=np.zeros(0), reg_values=None,
                         start_value=np.array([5.0, -3.0]))


@pytest.mark.parametrize("n,dim", [(3000, 256), (20011, 512)])
def test_xlx_matches_reference_expression(oracle, n, dim):
    """K6: X' (L / trace L) X (graph_based.py:45-49) -- f64 on both sides; the GPU differs from
    numpy's dgemm only in the order of the N-long f64 sums (tolerance 1e-11 of the matrix scale),
    far below the f32 cast MultiReg applies to the result (multi_reg.py:31)."""
    from seesaw_amd.knn_graph import get_weight_matrix, post_process_graph_df, rbf_kernel
    from seesaw_amd.loops.graph_based import compute_xlx
    import pandas as pd
    rng = np.random.default_rng(n)
    k = 10
    src = np.repeat(np.arange(n), k)
    dst = (src + rng.integers(1, n, size=src.shape[0])) % n
    df = post_process_graph_df(pd.DataFrame({"src_vertex": src.astype(np.int32), "dst_vertex": dst.astype(np.int32),
                                             "distance": rng.random(src.shape[0]).astype(np.float32) * 0.3}), nvec=n)
    L = get_weight_matrix(df, kfun=rbf_kernel(0.05), self_edges=False, normalized=False, symmetric=True, laplacian=True)
    X = oracle.synth_rows(3, 0, n, dim)
    got = compute_xlx(L, X)
    Ls = L / L.diagonal().sum()
    want = X.T @ (Ls @ X)  # the reference expression (f64: scipy upcasts X)
    assert got.shape == (dim, dim) and got.dtype == np.float64
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-11 * scale, np.abs(got - want).max() / scale
    assert np.array_equal(got.astype(np.float32), want.astype(np.float32)) or \
        np.abs(got.astype(np.float32) - want.astype(np.float32)).max() <= 2e-7 * scale


def test_resident_chaining_equals_host_round_trip(oracle):
    """set_prior + fit_resident + fetch / scores_to_index == fit_transform with the same prior as reg_values
    and start value (the ranking loop's call pattern), bit for bit; the f32 scores handed to the index
    equal `scores.astype(float32)` with labelled nodes at -inf."""
    from seesaw_amd.device_index import DeviceIndex
    from seesaw_amd.label_propagation import LabelPropagation
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    W = _W(g)
    n = W.shape[0]
    rng = np.random.default_rng(1)
    prior = rng.random(n)
    ids = rng.choice(n, size=40, replace=False).astype(np.int64)
    vals = (rng.random(40) > 0.5).astype(np.float64)
    lp = LabelPropagation(W, reg_lambda=1.0, max_iter=300)
    want = lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
    sweeps = lp.last_sweeps
    lp.set_prior(prior)
    for _ in range(2):  # twice: the installed prior survives a call
        lp.fit_resident(label_ids=ids, label_values=vals)
        assert lp.last_sweeps == sweeps
        assert np.array_equal(lp.fetch(), want)
    # a short list of nodes (PseudoLR's pseudo-labels, loops/util.py:19) without fetching the iterate
    pick = rng.choice(n, size=300, replace=True).astype(np.int64)
    assert np.array_equal(lp.gather(pick), want[pick])
    assert lp.gather(np.zeros(0, np.int64)).shape == (0,)
    with pytest.raises(RuntimeError):
        lp.gather(np.array([n], np.int64))
    dev = DeviceIndex.from_numpy(g["X"])
    lp.scores_to_index(dev, mask_labeled=True)
    s32 = want.astype(np.float32)
    s32[ids] = -np.inf
    rows, scores, _ = dev.topk(None, 25)
    order = np.lexsort((np.arange(n), -s32.astype(np.float64)))[:25]
    assert np.array_equal(rows, order) and np.array_equal(scores, s32[order])
    dev.close()
    lp.close()


@pytest.fixture
def tiny_slices(monkeypatch):
    """ssw_labelprop_create reads SSW_LP_SLICE_KB: 2 KB of f_old per column slice = 256 columns, so even the
    1500-node golden graph runs through the column-blocked sweep (6 passes per sweep)"""
    monkeypatch.setenv("SSW_LP_SLICE_KB", "2")


def test_column_blocked_sweep_bit_exact_vs_reference_golden(tiny_slices):
    """the multi-pass (column-sliced, carried running sum) form of the sweep: same f64 bits and the same sweep
    counts as the reference on all nine golden runs"""
    test_fit_transform_bit_exact_vs_reference_golden()
    test_ranker_matches_reference_golden()


@pytest.mark.parametrize("slice_kb", ["1", "8", "64"])
def test_column_blocked_equals_plain_sweep_on_a_large_graph(monkeypatch, oracle, slice_kb):
    """150 k nodes with a hub row and an empty row: blocked (various slice widths, up to 1172 slices) == plain, bit
    for bit, over 25 sweeps; spot rows against the CPU oracle"""
    from seesaw_amd.label_propagation import LabelPropagation
    n, deg = 150_000, 9
    rng = np.random.default_rng(3)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, n * deg)
    vals = rng.uniform(0.05, 1.0, n * deg)
    hub = rng.integers(0, n, 9000)
    keep = rows != 7
    rows = np.concatenate([rows[keep], np.full(9000, 5)])
    cols = np.concatenate([cols[keep], hub])
    vals = np.concatenate([vals[keep], rng.uniform(0.05, 1.0, 9000)])
    A = sp.coo_array((vals, (rows, cols)), shape=(n, n)).tocsr()
    W = (A + A.T).tocsr()
    W.sort_indices()
    prior = rng.uniform(0, 1, n)
    ids = rng.choice(n, 300, replace=False).astype(np.int64)
    lab = (rng.uniform(size=300) > 0.5).astype(np.float64)
    kw = dict(label_ids=ids, label_values=lab, reg_values=prior, start_value=prior)
    monkeypatch.delenv("SSW_LP_SLICE_KB", raising=False)
    plain = LabelPropagation(W, reg_lambda=1.0, max_iter=25, epsilon=-1.0)
    want = plain.fit_transform(**kw)
    plain.close()
    monkeypatch.setenv("SSW_LP_SLICE_KB", slice_kb)
    blocked = LabelPropagation(W, reg_lambda=1.0, max_iter=25, epsilon=-1.0)
    got = blocked.fit_transform(**kw)
    assert blocked.last_sweeps == 25
    blocked.close()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    ref = oracle.label_propagation(W, label_ids=ids, label_values=lab, reg_lambda=1.0, reg_values=prior,
                                   start_value=prior, max_iter=25, epsilon=-1.0)
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert np.array_equal(np.asarray(ref).view(np.uint64), got.view(np.uint64))


def _orders(W):
    """node orders to hold the permuted layout to: a random permutation (no structure at all), the reversal, and the
    reverse Cuthill-McKee order locality_order() would pick on a large clustered graph"""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    n = W.shape[0]
    rng = np.random.default_rng(n)
    rcm = np.empty(n, dtype=np.int32)
    pattern = sp.csr_matrix(W)
    rcm[np.asarray(reverse_cuthill_mckee(((pattern + pattern.T) != 0).tocsr(), symmetric_mode=True))] = np.arange(n, dtype=np.int32)
    return {"random": rng.permutation(n).astype(np.int32), "reversed": np.arange(n - 1, -1, -1, dtype=np.int32), "rcm": rcm}


@pytest.mark.parametrize("order", ["random", "reversed", "rcm"])
def test_locality_order_is_bit_exact_vs_reference_golden(order):
    """the graph stored in a node order of its own: rows keep their entries in ascending ORIGINAL column
    id, so every sweep adds the same products in the same order -- the reference's outputs and sweep counts, bit for bit"""
    from seesaw_amd.label_propagation import LabelPropagation
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    W = _W(g)
    perm = _orders(W)[order]
    for r in range(int(g["n_runs"])):
        lam = float(g[f"run{r}_lam"])
        start = g[f"run{r}_start"]
        lp = LabelPropagation(W, reg_lambda=lam, max_iter=300, node_order=perm)
        out = lp.fit_transform(label_ids=g[f"run{r}_ids"], label_values=g[f"run{r}_vals"],
                               reg_values=start if lam > 0 else None, start_value=start)
        assert lp.last_sweeps == int(g[f"run{r}_steps"]), (r, lp.last_sweeps)
        assert np.array_equal(out, g[f"run{r}_out"]), (r, np.abs(out - g[f"run{r}_out"]).max())
        lp.close()


@pytest.mark.parametrize("order", ["random", "rcm"])
def test_locality_order_through_every_entry_point(order):
    """resident chaining over a re-ordered graph: set_prior, fit_resident, fetch, gather, prior_as_result,
    device_scores (handed to the index's f64 re-scoring in ORIGINAL order) and scores_to_index all speak original ids"""
    import ctypes
    from seesaw_amd.device_index import DeviceIndex
    from seesaw_amd.label_propagation import LabelPropagation
    g = np.load(os.path.join(GOLDEN, "labelprop.npz"))
    W = _W(g)
    n = W.shape[0]
    rng = np.random.default_rng(3)
    prior = rng.random(n)
    ids = rng.choice(n, size=40, replace=False).astype(np.int64)
    vals = (rng.random(40) > 0.5).astype(np.float64)
    ref = LabelPropagation(W, reg_lambda=1.0, max_iter=300)
    want = ref.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
    lp = LabelPropagation(W, reg_lambda=1.0, max_iter=300, node_order=_orders(W)[order])
    lp.set_prior(prior)
    for _ in range(2):
        lp.fit_resident(label_ids=ids, label_values=vals)
        assert lp.last_sweeps == ref.last_sweeps
        assert np.array_equal(lp.fetch(), want)
    pick = rng.choice(n, size=300, replace=True).astype(np.int64)
    assert np.array_equal(lp.gather(pick), want[pick])
    # the device pointer other kernels index by original node id
    import torch
    ptr = lp.device_scores_ptr()
    t = torch.empty(n, dtype=torch.float64, device="cuda:0")
    hip = None
    for cand in (os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), "libamdhip64.so"):
        try:
            hip = ctypes.CDLL(cand)
            break
        except OSError:
            continue
    assert hip is not None
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(t.data_ptr(), ptr, n * 8, 3) == 0  # hipMemcpyDeviceToDevice
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), want)
    dev = DeviceIndex.from_numpy(g["X"])
    lp.scores_to_index(dev, mask_labeled=True)
    s32 = want.astype(np.float32)
    s32[ids] = -np.inf
    rows, scores, _ = dev.topk(None, 25)
    top = np.lexsort((np.arange(n), -s32.astype(np.float64)))[:25]
    assert np.array_equal(rows, top) and np.array_equal(scores, s32[top])
    # the rounds before the first negative label: the prior is the result, labelled nodes marked
    lp.prior_as_result(ids[:5])
    assert np.array_equal(lp.fetch(), prior)
    lp.scores_to_index(dev, mask_labeled=True)
    p32 = prior.astype(np.float32)
    p32[ids[:5]] = -np.inf
    rows, scores, _ = dev.topk(None, 25)
    top = np.lexsort((np.arange(n), -p32.astype(np.float64)))[:25]
    assert np.array_equal(rows, top) and np.array_equal(scores, p32[top])
    dev.close()
    lp.close()
    ref.close()


def test_locality_order_is_found_on_clustered_graphs_only():
    """locality_order(): reverse Cuthill-McKee pays on a k-NN graph of clustered vectors (neighbours share a cluster) and
    is declined on one of unclustered vectors (edges stay scattered whatever the order)"""
    from seesaw_amd.label_propagation import locality_order
    rng = np.random.default_rng(0)
    n, k, nc = 60000, 8, 300
    lab = rng.integers(0, nc, n)
    members = [np.nonzero(lab == c)[0] for c in range(nc)]
    rows = np.repeat(np.arange(n), k)
    clustered = np.concatenate([rng.choice(members[lab[i]], k) for i in range(n)])
    Wc = sp.csr_matrix((np.ones(n * k), (rows, clustered)), shape=(n, n))
    Wc = (Wc + Wc.T).tocsr()
    order = locality_order(Wc, min_nodes=1000, window=2048)
    assert order is not None and np.array_equal(np.sort(order), np.arange(n))
    near = np.abs(order[np.repeat(np.arange(n), np.diff(Wc.indptr))].astype(np.int64) - order[Wc.indices]) < 2048
    assert near.mean() > 0.9
    Wr = sp.csr_matrix((np.ones(n * k), (rows, rng.integers(0, n, n * k))), shape=(n, n))
    Wr = (Wr + Wr.T).tocsr()
    assert locality_order(Wr, min_nodes=1000, window=2048) is None
    assert locality_order(Wc) is None  # under 2^19 nodes the iterate stays in L2: not worth a permutation


def _session_graph(rng, n, deg, weight_hi):
    """a symmetric k-NN-like graph whose weights decide how many sweeps a propagation needs"""
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, n * deg)
    A = sp.coo_array((rng.uniform(0.02, weight_hi, n * deg), (rows, cols)), shape=(n, n)).tocsr()
    W = (A + A.T).tocsr()
    W.setdiag(0)
    W.eliminate_zeros()
    W.sum_duplicates()
    W.sort_indices()
    return W


@pytest.mark.parametrize("n,deg,weight_hi,ordered", [(20000, 6, 0.02, False), (20000, 6, 0.25, False), (70000, 5, 0.05, True),
                                                     (3000, 4, 0.6, False)])
def test_incremental_propagation_equals_full_sweeps(oracle, n, deg, weight_hi, ordered):
    """Round 5: consecutive fit_resident calls update the kept iterates instead of sweeping the whole graph
    (ssw_labelprop_last_run_info).  A session's worth of label changes -- a few added per round, one removed, one whose
    value flips, a round with no change, a round that touches a hub -- must give the CPU oracle's f64 bits and sweep
    count every round; the light-weight graphs converge in 2-3 sweeps (incremental from round 2 on), the heavy ones need
    more sweeps than are kept or reach n / 8 rows and fall back to the full sweeps, and both ways the answers are equal."""
    from seesaw_amd.label_propagation import LabelPropagation
    rng = np.random.default_rng(n + deg)
    W = _session_graph(rng, n, deg, weight_hi)
    prior = rng.uniform(0.05, 0.95, n)
    order = None
    if ordered:
        order = rng.permutation(n).astype(np.int32)
    lp = LabelPropagation(W, reg_lambda=1.0, max_iter=300, node_order=order)
    lp.set_prior(prior)
    labels = {}
    modes, kinds = [], []
    for rnd in range(9):
        if rnd == 4 and labels:
            labels.pop(sorted(labels)[len(labels) // 2])                # a label removed
        elif rnd == 5 and labels:
            k = sorted(labels)[0]
            labels[k] = 1.0 - labels[k]                                  # a label whose value changes
        elif rnd == 6:
            pass                                                         # nothing changes
        else:
            for v in rng.choice(n, size=int(rng.integers(1, 14)), replace=False):
                labels[int(v)] = float(rng.integers(0, 2))
        ids = np.array(sorted(labels), dtype=np.int64)
        vals = np.array([labels[int(i)] for i in ids], dtype=np.float64)
        ref, sweeps, conv = oracle.label_propagation(W, label_ids=ids, label_values=vals, reg_lambda=1.0, reg_values=prior,
                                                     start_value=prior, max_iter=300)
        lp.fit_resident(label_ids=ids, label_values=vals)
        modes.append(lp.last_incremental)
        kinds.append(lp.last_mode)
        got = lp.fetch()
        assert lp.last_sweeps == sweeps and lp.last_converged == conv, (rnd, lp.last_sweeps, sweeps)
        assert np.array_equal(got, ref), (rnd, lp.last_incremental, float(np.abs(got - ref).max()))
        pick = rng.choice(n, size=50).astype(np.int64)
        assert np.array_equal(lp.gather(pick), ref[pick])
    assert modes[0] is False
    if weight_hi <= 0.05:
        # few sweeps, small frontiers: rounds are updates -- except where a round's labels need a sweep MORE than the round
        # before kept (a session's first rounds, a flipped label), which continue with full sweeps from the kept iterates
        assert sum(modes[1:]) >= 2, modes
        if modes[-1]:
            assert lp.last_rows_recomputed < n // 4  # ... of a sliver of the graph
    print(f"n={n} deg={deg} w<={weight_hi}: sweeps {lp.last_sweeps}, run kinds per round {kinds} (0 full, 1 update, 2 update + full "
          f"sweeps from the kept iterates), rows in the last run {lp.last_rows_recomputed}")
    lp.close()


def test_incremental_state_is_dropped_when_it_must(oracle, monkeypatch):
    """a new prior, other parameters, a fit_transform in between, prior_as_result: each makes the next resident call a
    full one; SSW_LP_NO_INCREMENTAL=1 makes every call full; and the answers never change"""
    from seesaw_amd.label_propagation import LabelPropagation
    rng = np.random.default_rng(5)
    n = 12000
    W = _session_graph(rng, n, 5, 0.03)
    prior = rng.uniform(0.05, 0.95, n)
    ids = rng.choice(n, size=30, replace=False).astype(np.int64)
    vals = rng.integers(0, 2, 30).astype(np.float64)

    def check(lp, ids, vals, prior, want_incremental):
        ref, sweeps, _ = oracle.label_propagation(W, label_ids=ids, label_values=vals, reg_lambda=lp.reg_lambda, reg_values=prior,
                                                  start_value=prior, max_iter=300)
        lp.fit_resident(label_ids=ids, label_values=vals)
        assert lp.last_incremental is want_incremental
        assert lp.last_sweeps == sweeps and np.array_equal(lp.fetch(), ref)

    lp = LabelPropagation(W, reg_lambda=1.0, max_iter=300)
    lp.set_prior(prior)
    check(lp, ids[:10], vals[:10], prior, False)
    check(lp, ids[:12], vals[:12], prior, True)
    prior2 = rng.uniform(0.05, 0.95, n)
    lp.set_prior(prior2)                                   # a new text query
    check(lp, ids[:12], vals[:12], prior2, False)
    check(lp, ids[:14], vals[:14], prior2, True)
    lp.reg_lambda = 2.0                                    # another parameter
    check(lp, ids[:15], vals[:15], prior2, False)
    check(lp, ids[:16], vals[:16], prior2, True)
    lp.fit_transform(label_ids=ids[:5], label_values=vals[:5], reg_values=prior2, start_value=prior2)
    lp.set_prior(prior2)
    check(lp, ids[:17], vals[:17], prior2, False)
    lp.prior_as_result(ids[:3])
    check(lp, ids[:18], vals[:18], prior2, False)
    check(lp, ids[:20], vals[:20], prior2, True)
    lp.close()
    monkeypatch.setenv("SSW_LP_NO_INCREMENTAL", "1")       # read per call (round 6): every call runs the full sweeps
    lp = LabelPropagation(W, reg_lambda=1.0, max_iter=300)
    lp.set_prior(prior)
    check(lp, ids[:10], vals[:10], prior, False)
    check(lp, ids[:12], vals[:12], prior, False)
    assert lp.last_mode == 0
    monkeypatch.delenv("SSW_LP_NO_INCREMENTAL")
    check(lp, ids[:14], vals[:14], prior, False)           # (no transposed pattern was built under the switch ...)
    lp.set_prior(prior)                                    # ... the next session's set_prior builds it
    check(lp, ids[:15], vals[:15], prior, False)
    check(lp, ids[:16], vals[:16], prior, True)
    lp.close()


@pytest.mark.parametrize("weight_hi,no_inc", [(0.02, False), (1.5, False), (0.02, True)])
def test_label_list_growth_past_its_capacity_keeps_the_installed_labels(oracle, monkeypatch, weight_hi, no_inc):
    """the device-side label list starts with room for 1024 entries; growing it used to free the
    installed list before the full path cleared is_label[] by it (a clear over uninitialised ids: an out-of-bounds
    write, or stale clamps).  Labels go 1000 -> 1100 -> 2100 (two growths) with some removed on the way, on a graph
    whose propagation converges in 2-3 sweeps (incremental path), on one that needs >= 8 sweeps (full tracked path
    every call) and with the incremental path switched off: the oracle's bits every call."""
    from seesaw_amd.label_propagation import LabelPropagation
    if no_inc:
        monkeypatch.setenv("SSW_LP_NO_INCREMENTAL", "1")
    rng = np.random.default_rng(77)
    n = 30000
    W = _session_graph(rng, n, 5, weight_hi)
    prior = rng.uniform(0.05, 0.95, n)
    pool = rng.permutation(n).astype(np.int64)
    vals_all = rng.integers(0, 2, n).astype(np.float64)
    lp = LabelPropagation(W, reg_lambda=1.0, max_iter=300)
    lp.set_prior(prior)
    sweeps_seen = []
    for step, (lo, hi) in enumerate([(0, 1000), (0, 1020), (0, 1100), (40, 1100), (40, 2100), (1500, 2100), (1500, 2105)]):
        ids = np.sort(pool[lo:hi])
        vals = vals_all[ids]
        ref, sweeps, conv = oracle.label_propagation(W, label_ids=ids, label_values=vals, reg_lambda=1.0, reg_values=prior,
                                                     start_value=prior, max_iter=300)
        lp.fit_resident(label_ids=ids, label_values=vals)
        sweeps_seen.append(sweeps)
        assert lp.last_sweeps == sweeps and lp.last_converged == conv, (step, lp.last_sweeps, sweeps)
        assert np.array_equal(lp.fetch(), ref), (step, lp.last_mode)
        if no_inc:
            assert lp.last_mode == 0
    if weight_hi >= 1.0:
        assert min(sweeps_seen) >= 8, sweeps_seen          # more sweeps than are kept: the full tracked path every call
    # and the handle still serves the other entry points (the untracked run clears by the same list)
    ids = np.sort(pool[5000:5010])
    ref, sweeps, _ = oracle.label_propagation(W, label_ids=ids, label_values=vals_all[ids], reg_lambda=1.0, reg_values=prior,
                                              start_value=prior, max_iter=300)
    out = lp.fit_transform(label_ids=ids, label_values=vals_all[ids], reg_values=prior, start_value=prior)
    assert np.array_equal(out, ref)
    lp.close()


@pytest.mark.parametrize("max_iter", [1, 2, 3, 4])
def test_resident_runs_with_a_small_sweep_budget(oracle, max_iter):
    """with max_iter no larger than the kept levels an incremental pass that does not converge has
    nothing left to sweep; the call must still report max_iter sweeps, not converged, and return iterate max_iter as
    the reference does.  Whatever path each call takes, values / sweep counts / convergence are the oracle's."""
    from seesaw_amd.label_propagation import LabelPropagation
    rng = np.random.default_rng(100 + max_iter)
    n = 8000
    kinds = []
    for weight_hi in (0.02, 0.08):
        W = _session_graph(rng, n, 5, weight_hi)
        prior = rng.uniform(0.05, 0.95, n)
        lp = LabelPropagation(W, reg_lambda=1.0, max_iter=max_iter)
        lp.set_prior(prior)
        labels = {}
        for rnd in range(8):
            for v in rng.choice(n, size=int(rng.integers(1, 30)), replace=False):
                labels[int(v)] = float(rng.integers(0, 2))
            ids = np.array(sorted(labels), dtype=np.int64)
            vals = np.array([labels[int(i)] for i in ids], dtype=np.float64)
            ref, sweeps, conv = oracle.label_propagation(W, label_ids=ids, label_values=vals, reg_lambda=1.0, reg_values=prior,
                                                         start_value=prior, max_iter=max_iter)
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                lp.fit_resident(label_ids=ids, label_values=vals)
            kinds.append(lp.last_mode)
            assert (lp.last_sweeps, lp.last_converged) == (sweeps, conv), (rnd, lp.last_mode, lp.last_sweeps, sweeps, conv)
            assert np.array_equal(lp.fetch(), ref), (rnd, lp.last_mode)
        lp.close()
    print(f"max_iter={max_iter}: run kinds {kinds}")


@pytest.mark.parametrize("n_images,tiles,weight_hi,ordered", [(900, 5, 0.03, False), (9000, 9, 0.03, False), (9000, 9, 0.25, False),
                                                              (8000, 10, 0.04, True)])
def test_fused_round_equals_the_three_calls(oracle, n_images, tiles, weight_hi, ordered):
    """Round 6: ssw_labelprop_round = run_resident (or prior_as_result) + scores_to_index + topk(q = NULL) in one call
    and -- when the propagation is an incremental update -- one host wait.  Over a session (rounds without a negative
    label, the first full propagation, updates, a label removed, a big change that overflows the frontier) the images,
    scores and best rows are those of the three calls on a second pair of handles, the propagated f64 scores are the CPU
    oracle's bits, on the small index form (<= 65 536 rows: one selection launch) and the general one."""
    from seesaw_amd.device_index import DeviceIndex
    from seesaw_amd.label_propagation import LabelPropagation
    rng = np.random.default_rng(n_images + tiles)
    n = n_images * tiles
    W = _session_graph(rng, n, 5, weight_hi)
    prior = rng.uniform(0.05, 0.95, n)
    X = rng.standard_normal((n, 256)).astype(np.float32)
    row2image = np.repeat(np.arange(n_images, dtype=np.int32), tiles)
    order = rng.permutation(n).astype(np.int32) if ordered else None
    pairs = []
    for _ in range(2):
        idx = DeviceIndex.from_numpy(X, row2image=row2image)
        lp = LabelPropagation(W, reg_lambda=1.0, max_iter=300, node_order=order)
        lp.set_prior(prior)
        pairs.append((idx, lp))
    (idx_a, lp_a), (idx_b, lp_b) = pairs
    labels, excluded = {}, []
    kinds = []
    k = 50
    for rnd in range(10):
        if rnd == 6 and labels:
            labels.pop(sorted(labels)[1])
        elif rnd == 8:
            for v in rng.choice(n, size=n // 6, replace=False):      # a change that reaches more than n / 8 rows
                labels[int(v)] = float(rng.integers(0, 2))
        else:
            for v in rng.choice(n, size=int(rng.integers(1, 14)), replace=False):
                labels[int(v)] = 1.0 if rnd < 2 else float(rng.integers(0, 2))   # no negative label in the first two rounds
        excluded = sorted(set(excluded) | set(int(v) for v in rng.choice(n_images, size=3, replace=False)))
        ids = np.array(sorted(labels), dtype=np.int64)
        vals = np.array([labels[int(i)] for i in ids], dtype=np.float64)
        propagate = bool((vals == 0).any())
        ex = np.asarray(excluded, dtype=np.int64)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            got = lp_a.round(idx_a, propagate=propagate, label_ids=ids, label_values=vals, mask_labeled=True, excluded=ex, k=k)
            lp_a._read_run_info()
            kinds.append((lp_a.last_mode, lp_a.last_host_syncs))
            if propagate:
                lp_b.fit_resident(label_ids=ids, label_values=vals)
            else:
                lp_b.prior_as_result(ids)
            lp_b.scores_to_index(idx_b, mask_labeled=True)
            want = idx_b.topk(None, k, excluded=ex)
        for g, w, what in zip(got, want, ("images", "scores", "rows")):
            assert np.array_equal(g, w), (rnd, what, kinds[-1])
        if propagate:
            ref, sweeps, conv = oracle.label_propagation(W, label_ids=ids, label_values=vals, reg_lambda=1.0, reg_values=prior,
                                                         start_value=prior, max_iter=300)
            assert (lp_a.last_sweeps, lp_a.last_converged) == (sweeps, conv)
            assert np.array_equal(lp_a.fetch(), ref), (rnd, kinds[-1])
            # the selection is the reference's: unlabelled vectors, best tile per image, excluded images out, (score desc, image asc)
            s32 = ref.astype(np.float32)
            s32[ids] = -np.inf
            per_image = s32.reshape(n_images, tiles).max(1)
            per_image[ex] = -np.inf
            top = np.lexsort((np.arange(n_images), -per_image.astype(np.float64)))[:k]
            top = top[np.isfinite(per_image[top])]
            assert np.array_equal(got[0], top), rnd
        else:
            assert np.array_equal(lp_a.fetch(), prior)
    modes = [m for m, _ in kinds]
    assert modes[0] == 3 and modes[1] == 3                      # the prior served as the result
    assert 0 in modes                                            # a first full propagation
    if weight_hi <= 0.05:
        assert modes.count(1) >= 3, kinds                        # updates ...
        assert all(s == 1 for m, s in kinds if m == 1), kinds    # ... with ONE host wait each
    print(f"n={n} w<={weight_hi} ordered={ordered}: (run kind, host waits) per round {kinds}")
    for idx, lp in pairs:
        lp.close()
        idx.close()
