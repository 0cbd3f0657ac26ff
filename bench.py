#!/usr/bin/env python3
"""bench.py -- headline benchmark: brute-force cosine top-k over a resident N x 512 f32 index.

Metric (BASELINE.json): vectors scanned / second on the 100M x 512 top-k configuration.
A "step" is one query: scan all rows of the (row-sharded) index, select the local top-k,
all-gather the per-shard top-k over RCCL (N > 1 only) and merge.  The index and the
queries are resident in HBM before the timed region starts.

    python bench.py                         # 1 GPU, 100M rows (204.8 GB), defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Scaling is STRONG: the same 100M rows are split over the N ranks (BASELINE config C4).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s)
ROW_BYTES = 512 * 4    # algorithmic bytes per vector: the row is read exactly once


def synth_query(seed: int, dim: int = 512):
    """unit-norm query vector (inputs of the timed path are generated here, not by the oracle)"""
    import numpy as np
    q = np.random.default_rng(10_000 + seed).standard_normal(dim).astype(np.float32)
    return (q / np.linalg.norm(q)).astype(np.float32)


def cpu_baseline(local_index, k, sample_rows, n_queries):
    """The reference expression timed on this host's cores (numpy, all BLAS threads):
    scores = X @ q; np.argsort(-scores); first k distinct non-excluded images
    (seesaw/indices/multiscale/multiscale_index.py:170-199) on a bounded sample."""
    import numpy as np
    import torch
    from oracle import seesaw_oracle as orc

    n = min(sample_rows, local_index.n_rows)
    X = local_index.download(0, n)
    row_dbidx = np.arange(n, dtype=np.int64)
    qs = [orc.synth_query(1000 + i) for i in range(n_queries)]
    orc.topk_images_reference(X, qs[0], row_dbidx, None, k)  # warm-up
    t0 = time.perf_counter()
    for q in qs:
        orc.topk_images_reference(X, q, row_dbidx, None, k)
    dt = time.perf_counter() - t0

    def argpartition_topk(q):
        """the fairer CPU variant of SURVEY section 8d: no full sort -- np.argpartition for the k best, then an
        ordering of those k (one vector per image here, so the k best rows are the k best images)"""
        scores = X @ q
        part = np.argpartition(-scores, k)[:k]
        return part[np.argsort(-scores[part], kind="stable")]

    ref_ids = orc.topk_images_reference(X, qs[0], row_dbidx, None, k)[0]
    assert np.array_equal(np.sort(argpartition_topk(qs[0])), np.sort(np.asarray(ref_ids))), "argpartition variant disagrees"
    t0 = time.perf_counter()
    for q in qs:
        argpartition_topk(q)
    dt_part = time.perf_counter() - t0
    # the expression is numpy's: its thread count is the BLAS pool's (threadpoolctl), not torch's
    blas_threads, blas_lib = None, None
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas" and "numpy" in str(p.get("filepath", ""))] or \
                [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            blas_threads, blas_lib = int(pools[0]["num_threads"]), f"{pools[0].get('internal_api')} {pools[0].get('version')}"
    except Exception:
        pass
    return {
        "value": n * n_queries / dt,
        "unit": "vectors/s",
        "cores": int(blas_threads or torch.get_num_threads()),
        "blas_threads": blas_threads, "blas_library": blas_lib, "torch_threads": int(torch.get_num_threads()),
        "kind": "port",
        "sample_short": f"{n_queries} queries x {n} rows x 512 f32: X@q + np.argsort + distinct top-{k} (numpy)",
        "sample": f"{n_queries} queries x {n} rows x 512 f32 (first rows of the same index): "
                  f"X@q + np.argsort(-scores) + distinct-image top-{k}, numpy {np.__version__}, "
                  f"os.cpu_count()={os.cpu_count()}",
        "seconds": dt,
        "argpartition_variant": {"value": n * n_queries / dt_part, "unit": "vectors/s", "seconds": dt_part,
                                 "what": f"X@q + np.argpartition(-scores, {k}) + sort of the {k} kept: same rows, same queries"},
    }


class PhaseTimers:
    """Per-phase wall time of a feedback loop: the product's entry points are wrapped for the
    duration of one session -- index top-k (scan kernel by HIP events + selection/fetch), label propagation, the
    L-BFGS fit -- and what is left of the iteration latency is host code (session bookkeeping, pandas records)."""

    def __init__(self):
        self.t = {"topk_call": 0.0, "label_prop": 0.0, "fit": 0.0, "sample_draw": 0.0}
        # the label-propagation phase by entry point, and what the propagation calls did on the device
        self.lp = {"propagate": 0.0, "scores_to_index": 0.0, "fetch": 0.0, "calls": 0, "sweeps": 0, "launches": 0, "host_syncs": 0,
                   "incremental": 0, "fused": 0, "rows": 0, "frontier_us": 0.0, "device_wait_us": 0.0}
        self._saved = []

    def _wrap_lp(self, cls, name, key):
        orig = getattr(cls, name)
        timers, lp = self.t, self.lp

        timed_name = name

        def timed(obj, *a, **k):
            t0 = time.perf_counter()
            try:
                return orig(obj, *a, **k)
            finally:
                dt = time.perf_counter() - t0
                timers["label_prop"] += dt
                lp[key] += dt
                if key == "propagate":
                    lp["calls"] += 1
                    lp["sweeps"] += int(getattr(obj, "last_sweeps", 0))
                    lp["launches"] += int(getattr(obj, "last_launches", 0))
                    lp["host_syncs"] += int(getattr(obj, "last_host_syncs", 0))
                    lp["incremental"] += int(getattr(obj, "last_mode", 0) == 1)
                    lp["fused"] += int(timed_name == "round")
                    lp["rows"] += int(getattr(obj, "last_rows_recomputed", 0))
                    lp["frontier_us"] += float(getattr(obj, "last_frontier_us", 0.0))
                    lp["device_wait_us"] += float(getattr(obj, "last_device_wait_us", 0.0))

        self._saved.append((cls, name, orig))
        setattr(cls, name, timed)

    def _wrap(self, cls, name, key):
        orig = getattr(cls, name)
        timers = self.t

        def timed(obj, *a, **k):
            t0 = time.perf_counter()
            try:
                return orig(obj, *a, **k)
            finally:
                timers[key] += time.perf_counter() - t0

        self._saved.append((cls, name, orig))
        setattr(cls, name, timed)

    def _wrap_function(self, module, name, key):
        orig = getattr(module, name)
        timers = self.t

        def timed(*a, **k):
            t0 = time.perf_counter()
            try:
                return orig(*a, **k)
            finally:
                timers[key] += time.perf_counter() - t0

        self._saved.append((module, name, orig))
        setattr(module, name, timed)

    def __enter__(self):
        from seesaw_amd.loops import util as loops_util
        from seesaw_amd.device_index import DeviceIndex
        from seesaw_amd.label_propagation import LabelPropagation
        from seesaw_amd.logistic_regression import LogisticRegressionPT
        from seesaw_amd.loops.multi_reg import RegModule
        self._wrap(DeviceIndex, "topk", "topk_call")
        for m, key in (("fit_transform", "propagate"), ("fit_resident", "propagate"), ("round", "propagate"),
                       ("scores_to_index", "scores_to_index"), ("fetch", "fetch")):
            self._wrap_lp(LabelPropagation, m, key)
        self._saved.append((LabelPropagation, "collect_run_info", LabelPropagation.collect_run_info))
        LabelPropagation.collect_run_info = True
        self._wrap(RegModule, "fit", "fit")
        self._wrap(LogisticRegressionPT, "fit", "fit")
        self._wrap_function(loops_util, "permutation_prefix", "sample_draw")  # PseudoLR's np.random.permutation(n)[:k], drawn by the library
        return self

    def __exit__(self, *exc):
        for cls, name, orig in self._saved:
            setattr(cls, name, orig)
        self._saved = []

    def per_iteration_ms(self, dev_index, latencies):
        n = max(1, len(latencies))
        scan = dev_index.profile_read()
        total = 1e3 * float(sum(latencies)) / n
        scan_ms = float(scan.sum()) / n if len(scan) else 0.0
        topk = 1e3 * self.t["topk_call"] / n
        lp, fit, draw = 1e3 * self.t["label_prop"] / n, 1e3 * self.t["fit"] / n, 1e3 * self.t["sample_draw"] / n
        d = self.lp
        calls = max(1, d["calls"])
        lp_detail = {"propagate_ms": 1e3 * d["propagate"] / n, "scores_to_index_ms": 1e3 * d["scores_to_index"] / n,
                     "fetch_ms": 1e3 * d["fetch"] / n, "propagations": d["calls"], "sweeps_per_propagation": d["sweeps"] / calls,
                     "launches_per_propagation": d["launches"] / calls, "host_syncs_per_propagation": d["host_syncs"] / calls,
                     "incremental_propagations": d["incremental"], "fused_rounds": d["fused"],
                     "rows_recomputed_per_propagation": d["rows"] / calls,
                     "host_frontier_us_per_propagation": d["frontier_us"] / calls, "device_wait_us_per_propagation": d["device_wait_us"] / calls}
        return {"iteration": total, "scan_kernel": scan_ms, "select_and_fetch": max(0.0, topk - scan_ms),
                "label_prop": lp, "label_prop_detail": lp_detail if d["calls"] else None,
                "fit": fit, "sample_draw": draw, "host_other": max(0.0, total - topk - lp - fit - draw),
                "host_syncs_per_round": (d["host_syncs"] / calls) if d["calls"] else None,
                "note": "ms per iteration of the reported session; scan_kernel by HIP events around the scan launches, the "
                        "other phases by host wall time around the C-ABI calls (they synchronise); the timed session is a "
                        "third one, run after the reported one (the wrappers cost ~1 us per call)"}


def feedback_loop_extras(device: int, full_images: int, with_cpu: bool = True, clustered_graph: bool = False):
    """seesaw_bench feedback-loop iterations / s (1 / mean(latencies), seesaw_bench.py:310,352)
    on the LVIS-shape synthetic datasets (BASELINE config C5), HIP path next to the CPU oracle
    (numpy / scipy / torch-CPU, the reference's own expressions) in the same run.
    with_cpu=False (ranks of an N > 1 run: every GPU runs its own replica of the sessions, the way
    seesaw_bench's parallel_run spreads sessions over actors) skips the CPU side and the sweep timing."""
    import numpy as np
    import torch
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import make_session
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset

    matrix = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
    loops = {
        "plain": None,
        "multi_reg": dict(label_loss_type="ce_loss", rank_loss_margin=0.2, use_qvec_norm=None, reg_data_lambda=0.0,
                          reg_norm_lambda=100.0, reg_query_lambda=0.0, verbose=False, max_iter=200,
                          pos_weight="balanced", lr=1.0, matrix_options=matrix),
        "knn_prop2": dict(matrix_options=matrix, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0,
                          calib_b=-0.4, prior_weight=1.0),
        "pseudo_lr": dict(switch_over=True, real_sample_weight=1.0, sample_size=10000,
                          log_reg_params=dict(class_weights=1.0, scale="centered", reg_lambda=1.0, max_iter=200.0, lr=1,
                                              fit_intercept=False),
                          label_prop_params=dict(matrix_options=matrix, normalize_scores=False,
                                                 sigmoid_before_propagate=True, calib_a=10.0, calib_b=-0.4,
                                                 prior_weight=1.0)),
    }
    # "<loop>@avg": the same loop under the aggregation of the reference's standard bench config
    # (scripts/configs/std_bench.yaml: agg_method avg_score, aug_larger all) -- HIP side only
    loops["plain@avg"] = loops["plain"]
    loops["knn_prop2@avg"] = loops["knn_prop2"]
    cpu_legs = ("plain", "multi_reg", "knn_prop2", "pseudo_lr")  # loops oracle/cpu_loop.py restates
    out = {}
    import contextlib
    import io
    for tag, n_images, knn_k, names in (("lvis_1109x13", 1109, 10, ("plain", "multi_reg", "knn_prop2", "pseudo_lr")),
                                        (f"full_{full_images}x13", full_images, 10,
                                         ("plain", "multi_reg", "knn_prop2", "pseudo_lr", "plain@avg", "knn_prop2@avg"))):
        ds = make_dataset("lvis", n_images=n_images, tiles_per_image=13, n_categories=2, positive_frac=0.05,
                          seed=11, knn_k=knn_k, device=device)
        ds.embedding.noise = 1.2  # a mediocre text query, so the loop runs all its rounds
        gdm = GlobalDataManager().add(ds)
        boxes, _ = ds.load_ground_truth()
        res = {"vectors": int(ds.vectors.shape[0])}
        full = n_images > 20000
        t0 = time.perf_counter()
        ds.knn_graph()  # exact k-NN graph (k = 10) built on the GPU: ssw_knn_build
        t_graph = time.perf_counter() - t0
        nv = float(ds.vectors.shape[0])
        res["knn_graph"] = {"k": knn_k, "build_s_incl_upload_and_dataframe": t_graph,
                            "pair_scores": nv * nv, "exact": True}
        for name in names:
            avg = name.endswith("@avg")
            p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale"), interactive=name.split("@")[0],
                              interactive_options=loops[name], batch_size=1, shortlist_size=50,
                              agg_method="avg_score" if avg else "plain_score", aug_larger="all" if avg else "greater",
                              start_policy="after_first_batch", index_options={"use_vec_index": False})
            b = BenchParams(name=name, ground_truth_category="c1", qstr="a c1", n_batches=30, max_results=10 ** 6)
            with contextlib.redirect_stdout(io.StringIO()):
                ret = make_session(gdm, p, b=b)
                runs = []
                for _ in range(3):  # the first session warms kernels / allocations; the headline is the mean over the rounds
                    #                 of BOTH timed ones (each session's own mean beside it; median / slowest round of the
                    #                 faster session say whether a mean carries a one-off hiccup of the box)
                    ret = make_session(gdm, p, b=b)
                    np.random.seed(0)
                    torch.manual_seed(0)
                    g = benchmark_loop(session=ret["session"], box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
                    runs.append((float(np.mean(g["latencies"])), g, ret))
                _, g, ret = min(runs[1:], key=lambda r: r[0])
                hip_shown = [int(v) for a in ret["session"].acc_indices for v in np.asarray(a).reshape(-1)]
                both = np.concatenate([np.asarray(runs[1][1]["latencies"]), np.asarray(runs[2][1]["latencies"])])
                res[name] = {"hip_iters_per_s": 1.0 / float(np.mean(both)), "hip_ms_per_iter": 1e3 * float(np.mean(both)),
                             "protocol": "one warm-up session, then the mean over ALL rounds of two timed sessions (no selection); "
                                         "the CPU leg is one session, unselected as well",
                             "hip_ms_per_iter_each_session": [1e3 * runs[1][0], 1e3 * runs[2][0]], "iters": len(g["latencies"]),
                             # the metric is 1 / mean (seesaw_bench.py:310,352); the median and the slowest round say
                             # whether a mean carries a one-off hiccup of the box
                             "hip_ms_per_iter_median": 1e3 * float(np.median(g["latencies"])),
                             "hip_ms_slowest_iter": 1e3 * float(np.max(g["latencies"])),
                             "hip_nfound": g["nfound"]}
                if with_cpu:  # per-phase ms (rank 0 at N = 1 only)
                    ret = make_session(gdm, p, b=b)
                    dev = ds.load_index()._dev
                    dev.profile(True)
                    np.random.seed(0)
                    with PhaseTimers() as ph:
                        # every round of this session stamped (name what the slowest round spends): time inside
                        # session.next() / session.refine() and what the wrapped entry points took of it, round by round
                        sess, per_round, mark = ret["session"], [], {}
                        s_next, s_refine = sess.next, sess.refine

                        def next_stamped(*a, **k):
                            mark.clear()
                            mark.update(t0=time.perf_counter(), snap=dict(ph.t))
                            try:
                                return s_next(*a, **k)
                            finally:
                                mark["next_ms"] = 1e3 * (time.perf_counter() - mark["t0"])

                        def refine_stamped(*a, **k):
                            t1 = time.perf_counter()
                            try:
                                return s_refine(*a, **k)
                            finally:
                                t2 = time.perf_counter()
                                row = {"next_ms": mark.get("next_ms"), "bookkeeping_ms": 1e3 * (t1 - mark["t0"]) - mark.get("next_ms", 0.0),
                                       "refine_ms": 1e3 * (t2 - t1)}
                                row.update({k2 + "_ms": 1e3 * (ph.t[k2] - mark["snap"][k2]) for k2 in ph.t})
                                per_round.append({k2: round(v, 4) for k2, v in row.items()})
                        sess.next, sess.refine = next_stamped, refine_stamped
                        gp = benchmark_loop(session=sess, box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
                    res[name]["phases_ms"] = ph.per_iteration_ms(dev, gp["latencies"])
                    lat_ms = [round(1e3 * float(v), 4) for v in gp["latencies"]]
                    slow = int(np.argmax(lat_ms)) if lat_ms else 0
                    res[name]["rounds"] = {"latency_ms": lat_ms, "per_round": per_round, "slowest_round": slow + 1,
                                           "slowest_round_phases": per_round[slow] if slow < len(per_round) else None,
                                           "note": "the phase session's rounds (wrapped entry points): topk_call = scan + selection + fetch, "
                                                   "label_prop = the graph handle's calls (the fused round includes its selection)"}
                    dev.profile(False)
                if not with_cpu or name not in cpu_legs:
                    continue
                from oracle import cpu_loop  # the CPU leg: the reference's expressions on the host cores
                qvec = ds.load_index().string2vec("a c1")
                # bounded CPU sample (0.25-0.9 s a round at 1.56 M vectors, 0.35-0.4 s for the L-BFGS loops at 14 417): the
                # rounds compared here are a sample; all 30 rounds of every loop at the full size are compared in
                # tests/test_c5_fullsize_gpu.py, and the reference's own 30-round sessions at the small size in test_c5_sequence_gpu.py
                cpu_rounds = (9 if name in ("knn_prop2", "pseudo_lr") else 8) if full else (10 if name in ("multi_reg", "pseudo_lr") else 30)
                np.random.seed(0)      # both legs draw from numpy's / torch's global streams (box-drop draws, PseudoLR's
                torch.manual_seed(0)   # sample, nn.Linear start weights): same seeds, same draws
                c = cpu_loop.run_session(ds.vectors, ds.vector_meta, boxes, "c1", qvec, loop=name, n_batches=cpu_rounds,
                                         max_results=10 ** 6, knn_df=ds.knn_graph().restrict_k(k=10).knn_df if knn_k else None)
            # the two legs' image sequences, round by round (not only nfound): the CPU leg restates the reference's
            # expressions, so this is the C5 parity check at the stated size.  The L-BFGS loops may part where the
            # reference itself does not reproduce its fits; the agreement is reported, not assumed.
            cpu_shown = [int(v) for v in c["shown"]]
            m_cmp = min(len(cpu_shown), len(hip_shown))
            prefix = 0
            while prefix < m_cmp and cpu_shown[prefix] == hip_shown[prefix]:
                prefix += 1
            res[name].update({"cpu_iters_per_s": 1.0 / float(np.mean(c["latencies"])),
                              "cpu_ms_per_iter": 1e3 * float(np.mean(c["latencies"])),
                              "cpu_iters_timed": len(c["latencies"]), "cpu_nfound": c["nfound"],
                              "sequence_check": {"rounds_compared": m_cmp, "identical_prefix": prefix,
                                                 "same_images": sorted(cpu_shown[:m_cmp]) == sorted(hip_shown[:m_cmp])}})
        if full and with_cpu:  # one label-propagation sweep against its HBM/L2 stream (12 B per non-zero + 40 B per node)
            from seesaw_amd.knn_graph import get_weight_matrix, rbf_kernel
            from seesaw_amd.label_propagation import LabelPropagation
            with contextlib.redirect_stdout(io.StringIO()):
                knn_df = ds.knn_graph().restrict_k(k=10).knn_df
                # get_weight_matrix (knn_graph.py:31-104), the one-off O(nnz) pass between the graph and the sweeps:
                # assembled on the device (csrc/wmatrix.hip) next to the reference's scipy form on the host cores
                get_weight_matrix(knn_df, kfun=rbf_kernel(0.05), self_edges=False, normalized=False, symmetric=True, device=device)
                tw = time.perf_counter()
                W = get_weight_matrix(knn_df, kfun=rbf_kernel(0.05), self_edges=False, normalized=False, symmetric=True,
                                      device=device)
                t_wm_dev = time.perf_counter() - tw
                tw = time.perf_counter()
                W_host = get_weight_matrix(knn_df, kfun=rbf_kernel(0.05), self_edges=False, normalized=False, symmetric=True)
                t_wm_host = time.perf_counter() - tw
                res["weight_matrix"] = {"nodes": int(W.shape[0]), "nnz": int(W.nnz), "device_s": t_wm_dev, "host_scipy_s": t_wm_host,
                                        "identical_arrays": bool(np.array_equal(W.indptr, W_host.indptr) and
                                                                 np.array_equal(W.indices, W_host.indices) and
                                                                 np.array_equal(W.data, W_host.data)),
                                        "note": "device_s includes the host-side exp(), the uploads and the copy back"}
                del W_host
                lp = LabelPropagation(W, reg_lambda=1.0, max_iter=1, epsilon=-1.0, device=device)
                prior = np.full(W.shape[0], 0.5)
                ids, vals = np.arange(0, 1000, dtype=np.int64), (np.arange(1000) % 2).astype(np.float64)
                lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
                t1 = time.perf_counter()
                lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
                t1 = time.perf_counter() - t1
                lp.max_iter = 201
                t201 = time.perf_counter()
                lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
                t201 = time.perf_counter() - t201
                lp.close()
            sweep_s = (t201 - t1) / 200.0
            nbytes = 12.0 * W.nnz + 40.0 * W.shape[0]
            res["labelprop_sweep"] = {"graph": "k-NN graph of the benchmark's (unclustered) vectors: no locality to order by",
                                      "nodes": int(W.shape[0]), "nnz": int(W.nnz), "ms_per_sweep": 1e3 * sweep_s,
                                      "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / sweep_s / 1e9,
                                      "frac_of_hbm_peak": nbytes / sweep_s / 1e9 / HBM_PEAK_GBS}
            del W
            if clustered_graph:  # --clustered-graph: 20 s of set-up (an exact k-NN build over clustered vectors + the RCM order)
                try:
                    res["labelprop_sweep_clustered"] = labelprop_clustered_extras(device, int(ds.vectors.shape[0]))
                except Exception as e:
                    res["labelprop_sweep_clustered"] = {"error": f"{type(e).__name__}: {e}"}
            else:
                res["labelprop_sweep_clustered"] = {"skipped": "python bench.py --clustered-graph (profiles/r04_bench_100M_output.json holds "
                                                               "the round-4 figures: 0.132 ms a sweep = 0.38 of the HBM peak with the locality order)"}
        out[tag] = res
        idx = ds.load_index()
        idx._dev.close()
    return out


def labelprop_clustered_extras(device: int, n: int):
    """The same sweep on a graph WITH locality: the exact k-NN graph (k = 10) of n mixture-of-Gaussians
    vectors (2000 clusters, nodes in random order -- what CLIP vectors of a real dataset look like to the graph), through
    the path the loops take: label_propagation.locality_order (reverse Cuthill-McKee, taken when it raises the share of
    near-diagonal edges at least twofold) and ssw_labelprop_create_ordered.  Bit-identical output under any node order."""
    import contextlib
    import io
    import numpy as np
    import scipy.sparse as sp
    import torch
    from seesaw_amd.knn_graph import compute_exact_knn, get_weight_matrix, rbf_kernel
    from seesaw_amd.label_propagation import LabelPropagation, locality_order
    dev = torch.device("cuda", device)
    gen = torch.Generator(device=dev).manual_seed(0)
    dim, n_clusters = 512, 2000
    centres = torch.randn(n_clusters, dim, device=dev, generator=gen)
    lab = torch.randint(0, n_clusters, (n,), device=dev, generator=gen)
    X = np.empty((n, dim), dtype=np.float32)
    for a in range(0, n, 1 << 18):
        b = min(n, a + (1 << 18))
        x = centres[lab[a:b]] + 0.3 * torch.randn(b - a, dim, device=dev, generator=gen)
        X[a:b] = torch.nn.functional.normalize(x, dim=1).cpu().numpy()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        df = compute_exact_knn(X, 10)
    t_knn = time.perf_counter() - t0
    del X
    W = sp.csr_matrix(get_weight_matrix(df, kfun=rbf_kernel(0.05), self_edges=False, normalized=False, symmetric=True, device=device))
    W.sort_indices()
    t0 = time.perf_counter()
    order = locality_order(W)
    t_order = time.perf_counter() - t0
    out = {"graph": f"exact 10-NN graph of {n} mixture-of-Gaussians vectors ({n_clusters} clusters, random node order)",
           "nodes": int(W.shape[0]), "nnz": int(W.nnz), "knn_build_s": t_knn, "locality_order_s": t_order,
           "order_taken": order is not None}
    nbytes = 12.0 * W.nnz + 40.0 * W.shape[0]
    prior = np.full(W.shape[0], 0.5)
    ids, vals = np.arange(0, 1000, dtype=np.int64), (np.arange(1000) % 2).astype(np.float64)
    sums = {}
    for tag, node_order in (("as_given", None), ("locality_order", order)):
        if tag == "locality_order" and order is None:
            continue
        lp = LabelPropagation(W, reg_lambda=1.0, max_iter=1, epsilon=-1.0, device=device, node_order=node_order)
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(2):
                lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
            t1 = time.perf_counter()
            lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
            t1 = time.perf_counter() - t1
            lp.max_iter = 201
            t2 = time.perf_counter()
            res = lp.fit_transform(label_ids=ids, label_values=vals, reg_values=prior, start_value=prior)
            t2 = time.perf_counter() - t2
        lp.close()
        sweep_s = (t2 - t1) / 200.0
        sums[tag] = np.asarray(res).tobytes()
        out[tag] = {"ms_per_sweep": 1e3 * sweep_s, "achieved_GBps": nbytes / sweep_s / 1e9,
                    "frac_of_hbm_peak": nbytes / sweep_s / 1e9 / HBM_PEAK_GBS}
    out["algorithmic_bytes"] = nbytes
    if len(sums) == 2:
        out["identical_output_under_both_orders"] = sums["as_given"] == sums["locality_order"]
    return out


def clip_extras(device: int):
    """CLIP ViT-B/32 (random-init weights, bf16 MFMA): tiles / s and achieved fraction of the
    dense bf16 MFMA peak (BASELINE config C3; 8.818 GFLOP per tile)."""
    import numpy as np
    import torch
    from seesaw_amd.models.clip import ClipModel
    m = ClipModel.random_init(seed=1234, device=device)
    dev = torch.device("cuda", device)
    B = 200
    x = torch.randn(B, 3, 224, 224, device=dev)
    o = torch.empty(B, 512, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    n, REPS = 10, 3

    def median_of(fn, loops=n):
        """median over REPS timed loops of `loops` calls each (2 untimed calls first); every loop's mean is kept"""
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        runs = []
        for _ in range(REPS):
            t0 = time.perf_counter()
            for _ in range(loops):
                fn()
            torch.cuda.synchronize(dev)
            runs.append((time.perf_counter() - t0) / loops)
        return float(np.median(runs)), [r * 1e3 for r in runs]

    def fwd():
        m.embed_image_dev(x.data_ptr(), B, o.data_ptr(), True, s)

    # 8.818 GFLOP is a tile's forward as transformers runs it.  The default form runs the last layer for the pooled row of
    # a tile only (the other 49 rows feed nothing: DESIGN section 4): the last attention for row 0's query, out-projection,
    # fc1 and fc2 on B rows -- flops that are NOT executed are not counted as achieved: tflops / frac_of_bf16_dense_peak are
    # on executed flops.  Every figure below is the MEDIAN of three timed loops of ten forwards (all three are listed).
    att_row = 2 * 2 * 50 * 768               # QK^T + PV of one query row against 50 keys, all heads
    GF_FULL = 8.818
    GF_RUN = GF_FULL - 49 * (4 * 768 * 3072 + 2 * 768 * 768 + att_row) / 1e9
    m.set_option(m.OPT_FULL_LAST_LAYER, True)  # every row through the last layer, as the reference's model computes it
    dtf, runs_f = median_of(fwd)
    m.set_option(m.OPT_FULL_LAST_LAYER, False)
    dt, runs_d = median_of(fwd)
    tf = B * GF_RUN / dt / 1e3
    # the same tower fed more tiles a call (not the C3 shape; reported beside it): 10 000 token rows leave the 256-row
    # tile kernels 1.4 and 1.9 rounds of the chip, 20 000 fill it -- what B = 200 loses is tile-count rounding.  1024 is
    # what the host entry points and the ingest tool hand over at a time (a tile's vector does not depend on its call).
    bigger = {}
    for B2 in (400, 1024):
        x2 = torch.randn(B2, 3, 224, 224, device=dev)
        o2 = torch.empty(B2, 512, device=dev)
        dt2, runs2 = median_of(lambda: m.embed_image_dev(x2.data_ptr(), B2, o2.data_ptr(), True, s), loops=5)
        del x2, o2
        bigger[B2] = {"ms_per_batch": dt2 * 1e3, "ms_per_batch_runs": runs2, "tiles_per_s": B2 / dt2, "tflops": B2 * GF_RUN / dt2 / 1e3,
                      "frac_of_bf16_dense_peak": B2 * GF_RUN / dt2 / 1e3 / 2500.0}
    # the box's floor for a chain of dependent launches (state it next to the text figures, which are
    # launch-bound: 6 launches a layer at 16 x 77, 4 for one short query): 200 one-element torch adds captured in a graph
    # and replayed -- no host in the loop, nothing to compute, so the time per node is what a launch boundary costs here
    floor_us = None
    try:
        z = torch.zeros(1, device=dev)
        side = torch.cuda.Stream(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            z.add_(1.0)
            torch.cuda.synchronize(dev)
            g.capture_begin()
            for _ in range(200):
                z.add_(1.0)
            g.capture_end()
        torch.cuda.synchronize(dev)
        g.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize(dev)
        floor_us = (time.perf_counter() - t0) / 5 / 200 * 1e6
        del g
    except Exception as e:  # the figure is context, not a result
        floor_us = None
        sys.stderr.write(f"launch floor probe failed: {type(e).__name__}: {e}\n")
    ids = np.random.default_rng(0).integers(0, 49405, (16, 77)).astype(np.int32)
    ids[:, 0], ids[:, -1] = 49406, 49407
    dtt, runs_t = median_of(lambda: m.embed_text(ids))
    one = ids[:1, :8].copy()  # one short query string, the interactive case (set_text)
    one[:, -1] = 49407
    dt1, runs_1 = median_of(lambda: m.embed_text(one), loops=20)
    # the other residual-row form (ssw_clip_set_option): bf16 rows -- faster, 4x the score error; the
    # headline figures above are the default f32 rows
    m.set_rows(image_bf16=True)
    dtb, runs_b = median_of(fwd)
    m.set_rows()
    m.close()
    # CPU side by side: the in-container transformers.CLIPModel, f32, torch-CPU,
    # all host threads, same random-init weights, on a bounded sample of 64 tiles, median of 3 runs
    import transformers
    torch.manual_seed(1234)
    hf = transformers.CLIPModel(transformers.CLIPConfig()).eval()
    xc = torch.randn(64, 3, 224, 224)
    with torch.inference_mode():
        hf.get_image_features(pixel_values=xc[:2])
        cpu_runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            hf.get_image_features(pixel_values=xc)
            cpu_runs.append(time.perf_counter() - t0)
    dtc = float(np.median(cpu_runs))
    cpu = {"tiles_per_s": 64 / dtc, "sample": "64 tiles a call, transformers.CLIPModel f32, torch-CPU, median of 3 calls",
           "threads": torch.get_num_threads(), "seconds": dtc, "seconds_runs": cpu_runs}
    return {"cpu_baseline": cpu, "timing": "median of 3 (three timed loops of ten forwards each; *_runs list all three, ms)",
            "image_batch": B, "image_ms_per_batch": dt * 1e3, "image_ms_per_batch_runs": runs_d, "tiles_per_s": B / dt, "tflops": tf,
            "mfma_peak_tflops": 2500.0, "frac_of_bf16_dense_peak": tf / 2500.0,
            # the same time priced at the reference model's flops per tile (8.818 GFLOP: what a user of the model gets per
            # second, the usual 'model flops utilisation'); the figure above counts only flops that were executed
            "model_flops_tflops": B * GF_FULL / dt / 1e3, "model_flops_frac_of_bf16_dense_peak": B * GF_FULL / dt / 1e3 / 2500.0,
            "gflop_per_tile": {"executed": GF_RUN, "full_model": GF_FULL,
                               "note": "default: the last layer (attention for row 0's query, out-projection, fc1, fc2) on the pooled row of a tile only; tflops count executed flops"},
            "image_full_last_layer": {"ms_per_batch": dtf * 1e3, "ms_per_batch_runs": runs_f, "tiles_per_s": B / dtf, "tflops": B * GF_FULL / dtf / 1e3,
                                      "frac_of_bf16_dense_peak": B * GF_FULL / dtf / 1e3 / 2500.0,
                                      "note": "SSW_CLIP_OPT_FULL_LAST_LAYER: every row through the last layer as the reference's model runs it (the default's vectors differ by 2e-5 ... 5e-5 on unit vectors; test bar 1e-4, tests/test_clip_gpu.py)"},
            "residual_rows": "f32 in both towers (the default; SURVEY 8 a-12's arithmetic)",
            "image_bf16_rows": {"ms_per_batch": dtb * 1e3, "ms_per_batch_runs": runs_b, "tiles_per_s": B / dtb, "tflops": B * GF_RUN / dtb / 1e3,
                                "frac_of_bf16_dense_peak": B * GF_RUN / dtb / 1e3 / 2500.0,
                                "note": "ssw_clip_set_option(SSW_CLIP_OPT_IMAGE_ROWS_BF16): max |score delta| 1.5e-3 against 4e-4 with f32 rows"},
            "image_batch_400": bigger[400], "image_batch_1024": bigger[1024],
            "text_batch": 16, "text_len": 77, "text_ms_per_batch_host_io": dtt * 1e3, "text_ms_per_batch_runs": runs_t,
            "texts_per_s": 16 / dtt, "text_tflops": 16 * 5.96 / dtt / 1e3, "text_frac_of_bf16_dense_peak": 16 * 5.96 / dtt / 1e3 / 2500.0,
            "single_query_8_tokens_ms_host_io": dt1 * 1e3, "single_query_8_tokens_ms_runs": runs_1,
            "launch_floor_us": floor_us,
            "launch_floor_note": "per node of a replayed graph of 200 dependent one-element kernels on this box; the text tower is "
                                 "12 layers x 6 launches (16 x 77) / 12 x 4 (one short query) + 4",
            "weights": "transformers.CLIPModel(CLIPConfig()) random init, seed 1234", "dtype": "bf16 MFMA, f32 accumulate"}


def c2_extras(device: int):
    """BASELINE config C2: 1 M x 512 on one GPU -- scan kernel bandwidth against the HBM peak and the latency
    of a complete host-to-host query (2-KB query in, packed top-100 out), 50 timed queries after 5 warm-ups,
    without and with 1000 excluded ids."""
    import numpy as np
    from seesaw_amd.device_index import DeviceIndex
    n = 1_000_000
    idx = DeviceIndex.synthetic(n, 512, seed=2024, device=device)
    qs = [synth_query(1000 + i) for i in range(55)]
    excluded = np.random.default_rng(5).choice(n, size=1000, replace=False).tolist()
    out = {"rows": n, "k": 100}
    for tag, ex in (("no_exclusion", None), ("excluded_1000", excluded)):
        for q in qs[:5]:
            idx.topk(q, 100, excluded=ex)
        idx.profile(True)
        t0 = time.perf_counter()
        for q in qs[5:]:
            idx.topk(q, 100, excluded=ex)
        wall = (time.perf_counter() - t0) / 50
        ms = idx.profile_read()
        idx.profile(False)
        gbs = n * ROW_BYTES / (float(np.mean(ms)) * 1e-3) / 1e9
        out[tag] = {"query_ms_host_to_host": wall * 1e3, "vectors_per_s": n / wall, "scan_kernel_ms": float(np.mean(ms)),
                    "scan_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
    idx.close()
    return out


def sharded_step_extras(device: int, k: int = 100):
    """What one rank of the 8-GPU configuration (C4: 12.5 M rows = 25.6 GB per GPU) does per query, MEASURED on one GPU
    (VERDICT r3 #5c): scan, selection writing the rank's exchange message, the exchange, the merge of eight messages.  The
    exchange itself cannot be measured on one GPU; it is replaced (a) by a device copy of the rank's own message into the
    eight slots of the gathered buffer -- the merge then does the work of the real run -- and (b) by the library's own
    RCCL entry point at world size 1 (ssw_topk_allgather, SSW_C_COMM=1's path, VERDICT r3 #5d), which prices the
    collective's launch without a peer.  overhead = step - scan kernel."""
    import ctypes
    import numpy as np
    import torch
    from seesaw_amd import _lib
    from seesaw_amd.device_index import DeviceIndex, decode_keys
    from seesaw_amd.sharded import ShardedTopK
    n, world = 12_500_000, 8
    dev = torch.device("cuda", device)
    free_b, _ = torch.cuda.mem_get_info(dev)
    if free_b < n * ROW_BYTES + (2 << 30):  # (the headline index is closed before the extras; another tenant may hold HBM)
        return {"skipped": f"{free_b / 1e9:.1f} GB of HBM free, the 12.5 M-row shard needs {n * ROW_BYTES / 1e9:.1f} GB"}
    idx = DeviceIndex.synthetic(n, 512, seed=2024, device=device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    idx.set_stream(stream)
    out = {"rows": n, "world_emulated": world}
    qs = torch.from_numpy(np.stack([synth_query(i) for i in range(30)])).to(dev)
    for kk in (k, 1024):
        x = ShardedTopK(rank=0, world=world, device=dev, image_offset=0, k_max=max(128, kk), with_best=False)
        x.attach(idx)

        def step(i, comm):
            idx.topk_dev(qs[i].data_ptr(), kk)
            if comm:   # RCCL all-gather of the message at world size 1 (lands in slot 0), then the other seven slots
                _lib.call("ssw_topk_allgather", x._comm, ctypes.c_void_p(stream), ctypes.c_void_p(x.send_buf.data_ptr()),
                          ctypes.c_void_p(x.all_buf.data_ptr()), x.msg_len)
                x.all_buf[1:] = x.send_buf
            else:
                x.all_buf[:] = x.send_buf
            _lib.call("ssw_topk_merge_msgs_dev", device, ctypes.c_void_p(stream), ctypes.c_void_p(x.all_buf.data_ptr()), world,
                      x.k_max, 0, kk, ctypes.c_void_p(x.out_keys.data_ptr()), ctypes.c_void_p(x.out_count.data_ptr()),
                      ctypes.c_void_p(x.flags.data_ptr()), ctypes.c_void_p(x.flags_seen.data_ptr()))

        res = {}
        for comm in ((False, True) if kk == k else (False,)):  # (creating the one-rank communicator takes seconds: once)
            if comm:
                world_keep = x.world
                try:
                    x.world = 1   # the communicator has one rank
                    x.use_c_comm()
                except Exception as e:
                    res["c_comm"] = {"error": f"{type(e).__name__}: {e}"}
                    continue
                finally:
                    x.world = world_keep
            for i in range(5):
                step(i, comm)
            torch.cuda.synchronize(dev)
            idx.profile(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5, 30):
                step(i, comm)
            e1.record()
            torch.cuda.synchronize(dev)
            scan_ms = float(np.mean(idx.profile_read()))
            idx.profile(False)
            step_ms = e0.elapsed_time(e1) / 25
            x.assert_no_overflow_seen()
            c = int(x.out_count.item())
            imgs, _ = decode_keys(x.out_keys[:c].cpu().numpy().view(np.uint64))
            res["c_comm" if comm else "device_copy"] = {
                "step_ms": step_ms, "scan_kernel_ms": scan_ms, "overhead_ms": step_ms - scan_ms,
                "scan_GBps": n * ROW_BYTES / (scan_ms * 1e-3) / 1e9, "merged_count": c,
                "distinct_after_merge_of_8_copies": int(np.unique(imgs).shape[0])}
        x.close_c_comm()
        out[f"k{kk}"] = res
    o = out[f"k{k}"].get("device_copy", {})
    if o:
        out["predicted_8gpu_vectors_per_s"] = 8 * n / ((o["step_ms"] + 0.02) * 1e-3)  # + ~20 us for the xGMI all-gather (not measurable here)
    idx.close()
    return out


def fit_extras(device: int):
    """the feedback update alone (MultiReg ce_loss objective, 200 L-BFGS iterations allowed) on labelled sets the size
    a session reaches: the whole step(closure) as one kernel launch against the same fit driven from the host one
    closure evaluation at a time (bit-identical results: tests/test_feedback_gpu.py), ms per fit over 20 fits"""
    import numpy as np
    from seesaw_amd import _lib
    from seesaw_amd.feedback import FbObjective, FeedbackEngine
    rng = np.random.default_rng(0)
    out = {}
    for n in (52, 208, 390):
        X = rng.standard_normal((n, 512)).astype(np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        q = X[:5].mean(0)
        y = (X @ q > np.quantile(X @ q, 0.8)).astype(np.float64)
        eng = FeedbackEngine(512, device=device)
        eng.set_query(q / np.linalg.norm(q))
        eng.set_data(X, center=True)
        eng.set_targets(y, np.ones(n))
        obj = FbObjective(kind=_lib.SSW_FB_MULTIREG, loss_type=_lib.SSW_FB_LOSS_CE, fit_intercept=0, reg_kind=0,
                          pos_weight=-1.0, reg_weight=0.0, margin=0.0, reg_norm_lambda=100.0, reg_data_lambda=0.0,
                          reg_query_lambda=0.0)
        w0 = (q / np.linalg.norm(q)).astype(np.float32)
        row = {}
        for tag, host in (("one_launch", False), ("host_driven", True)):
            if host:
                os.environ["SSW_FB_HOST_DRIVER"] = "1"
            try:
                eng.fit(obj, w0, 200)
                t0 = time.perf_counter()
                for _ in range(20):
                    _, info = eng.fit(obj, w0, 200)
                row[tag + "_ms"] = (time.perf_counter() - t0) / 20 * 1e3
            finally:
                os.environ.pop("SSW_FB_HOST_DRIVER", None)
            assert info["on_device"] is (not host)
        row.update(closure_evaluations=info["func_evals"], lbfgs_iterations=info["n_iter"])
        out[f"rows_{n}"] = row
        eng.close()
    return out


SCAN_KERNEL = "scan_scores_kernel<2,2,nt>"          # what launch_scan runs by default (scan.hip)
SCAN_SOURCES = ("seesaw_amd/csrc/scan.hip",)


def scan_source_sha256() -> str:
    """identity of the scan kernel's source: profiles/traffic.json is only valid for the tree it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for rel in SCAN_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def measured_traffic(rows_per_launch: int):
    """HBM bytes per scan launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the bench).
    profiles/traffic.json holds one record per launch shape (100 M rows on one GPU; 50 / 25 / 12.5 M rows = what one rank
    of a 2 / 4 / 8-GPU run scans); a record is used only when it was taken on THIS kernel source, variant and launch
    shape; otherwise null + the reason"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
    except Exception as e:
        return None, f"no usable profiles/traffic.json ({type(e).__name__})"
    if tj.get("kernel") != SCAN_KERNEL:
        return None, f"traffic.json is for kernel {tj.get('kernel')!r}, the bench ran {SCAN_KERNEL!r}"
    if tj.get("kernel_source_sha256") != scan_source_sha256():
        return None, "the scan kernel's source changed since profiles/traffic.json was measured (re-run tools/collect_profiles.sh)"
    rec = (tj.get("shapes") or {}).get(str(int(rows_per_launch)))
    if rec is None:
        have = sorted(int(k) for k in (tj.get("shapes") or {}))
        return None, f"traffic.json holds launch shapes {have}, this run has {rows_per_launch} rows per launch"
    return rec.get("hbm_bytes_per_launch"), (f"PMC passes at {rows_per_launch} rows per launch, git {tj.get('git_head', '?')[:12]}, "
                                            "same kernel source (sha256 match)")


def aggregate_replicas(replicas, world: int):
    """N > 1: sum of the per-GPU feedback-loop rates (every rank ran its own replica of the sessions)."""
    agg = {}
    for rep in replicas:
        for tag, res in (rep or {}).items():
            if not isinstance(res, dict):
                continue
            for name, v in res.items():
                if isinstance(v, dict) and "hip_iters_per_s" in v:
                    a = agg.setdefault(tag, {}).setdefault(name, {"iters_per_s_all_gpus": 0.0, "per_gpu": []})
                    a["iters_per_s_all_gpus"] += v["hip_iters_per_s"]
                    a["per_gpu"].append(v["hip_iters_per_s"])
    return {"scaling": "replicas only (no data-path collective)", "gpus": world, "aggregate": agg,
            "errors": [r["error"] for r in replicas if isinstance(r, dict) and "error" in r]}


LINE_LIMIT = 6000   # characters of the ONE stdout line (the driver keeps 8000 characters of stdout: VERDICT r5 #1)


def _num(v, digits=5):
    """numbers of the machine line: `digits` significant digits, ints and bools as they are, non-finite -> null"""
    import math
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    try:
        f = float(v)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(f):
        return None
    return float(f"{f:.{digits}g}")


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(full: dict) -> str:
    """The ONE stdout line, built from the full result (which goes to bench_detail.json and stderr): the contract's
    fields, `roofline`, `cpu_baseline`, `allgather_us`, and a flat `extras` of NUMBERS only -- one or two scalars per
    BASELINE config.  No per-session lists, no prose.  Asserted < LINE_LIMIT characters (tests/test_bench_cpu.py)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: (_num(full.get(k), 7) if k in ("value", "ms_per_step") else full.get(k)) for k in keep}
    cfg = full.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in ("workload", "rows_total", "dim", "k", "rows_per_gpu", "parallelism", "rccl_ranks", "collective")}
    r = full.get("roofline") or {}
    out["roofline"] = {"bound": r.get("bound"), "kernel": r.get("kernel"), "achieved": _num(r.get("achieved"), 6), "peak": r.get("peak"),
                       "unit": r.get("unit"), "frac": _num(r.get("frac")), "traffic": _num(r.get("traffic"), 7),
                       "avg_launch_ms": _num(r.get("avg_launch_ms"), 6), "launches": r.get("launches"),
                       "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch")}
    c = full.get("cpu_baseline")
    out["cpu_baseline"] = None if not c else {"value": _num(c.get("value"), 6), "unit": c.get("unit"), "cores": c.get("cores"),
                                              "kind": c.get("kind"), "sample": c.get("sample_short") or str(c.get("sample"))[:120],
                                              "blas_threads": c.get("blas_threads"),
                                              "argpartition_value": _num(_get(c, "argpartition_variant", "value"), 6)}
    t1 = full.get("top1") or {}
    out["top1"] = {"image": t1.get("image"), "score": t1.get("score")}  # the last query's best image: the N = 1 / N > 1 sanity anchor
    a = full.get("allgather_us")
    out["allgather_us"] = None if not a else {k: _num(a.get(k)) for k in ("mean", "median", "max", "steps", "bytes_per_rank")}
    ex = full.get("extras") or {}
    flat = {}

    def put(key, v, digits=4):
        v = _num(v, digits)
        if v is not None:
            flat[key] = v

    put("c2_1M_frac", _get(ex, "c2_one_million_rows", "no_exclusion", "frac_of_hbm_peak"))
    put("c2_1M_query_ms", _get(ex, "c2_one_million_rows", "no_exclusion", "query_ms_host_to_host"))
    put("c2_1M_excl_query_ms", _get(ex, "c2_one_million_rows", "excluded_1000", "query_ms_host_to_host"))
    put("c4_rank_step_ms", _get(ex, "sharded_step_12p5M_rows", "k100", "device_copy", "step_ms"))
    put("c4_rank_scan_ms", _get(ex, "sharded_step_12p5M_rows", "k100", "device_copy", "scan_kernel_ms"))
    put("c4_predicted_8gpu_vectors_per_s", _get(ex, "sharded_step_12p5M_rows", "predicted_8gpu_vectors_per_s"))
    cl = ex.get("clip") or {}
    put("c3_image_b200_ms", cl.get("image_ms_per_batch"))
    put("c3_image_frac_model_flops", cl.get("model_flops_frac_of_bf16_dense_peak"))
    put("c3_image_frac_executed", cl.get("frac_of_bf16_dense_peak"))
    put("c3_image_full_last_layer_ms", _get(cl, "image_full_last_layer", "ms_per_batch"))
    put("c3_image_b1024_frac", _get(cl, "image_batch_1024", "frac_of_bf16_dense_peak"))
    put("c3_text_16x77_ms", cl.get("text_ms_per_batch_host_io"))
    put("c3_text_frac", cl.get("text_frac_of_bf16_dense_peak"))
    put("c3_text_1x8_ms", cl.get("single_query_8_tokens_ms_host_io"))
    put("c3_launch_floor_us", cl.get("launch_floor_us"), 3)
    put("c3_cpu_tiles_per_s", _get(cl, "cpu_baseline", "tiles_per_s"))
    put("c3_gpu_tiles_per_s", cl.get("tiles_per_s"))
    fl = ex.get("feedback_loop") or {}
    for tag, res in fl.items():
        if not isinstance(res, dict):
            continue
        short = "c5s" if tag.startswith("lvis") else "c5"
        for name, v in res.items():
            if not (isinstance(v, dict) and "hip_iters_per_s" in v):
                continue
            nm = f"{short}_{name.replace('@', '_')}"
            put(nm + "_hip_iters_per_s", v.get("hip_iters_per_s"))
            put(nm + "_cpu_iters_per_s", v.get("cpu_iters_per_s"))
            if short == "c5":
                put(nm + "_ms_median", v.get("hip_ms_per_iter_median"))
                put(nm + "_ms_slowest", v.get("hip_ms_slowest_iter"))
                put(nm + "_slowest_round", _get(v, "rounds", "slowest_round"))
                put(nm + "_identical_prefix", _get(v, "sequence_check", "identical_prefix"))
                ph = v.get("phases_ms") or {}
                for k_src, k_dst in (("label_prop", "label_prop_ms"), ("fit", "fit_ms"), ("sample_draw", "draw_ms"), ("host_other", "host_ms")):
                    if ph.get(k_src):
                        put(f"{nm}_{k_dst}", ph.get(k_src), 3)
                put(nm + "_host_syncs_per_round", ph.get("host_syncs_per_round"), 3)
        put(short + "_sweep_frac", _get(res, "labelprop_sweep", "frac_of_hbm_peak"))
        put(short + "_sweep_ms", _get(res, "labelprop_sweep", "ms_per_sweep"))
    rep = ex.get("feedback_loop_replicas")
    if rep:
        for tag, names in (rep.get("aggregate") or {}).items():
            short = "c5s" if tag.startswith("lvis") else "c5"
            for name, v in names.items():
                put(f"{short}_{name.replace('@', '_')}_iters_per_s_all_gpus", v.get("iters_per_s_all_gpus"))
        flat["replica_errors"] = len(rep.get("errors") or [])
    errs = [k for k, v in ex.items() if isinstance(v, dict) and "error" in v]
    if errs:
        flat["sections_failed"] = len(errs)
    put("wall_s_extras", sum((ex.get("section_seconds") or {}).values()) if ex.get("section_seconds") else None)
    out["extras"] = flat
    out["detail"] = "bench_detail.json"
    line = json.dumps(out, separators=(",", ":"), allow_nan=False)
    if len(line) >= LINE_LIMIT:  # never let the line outgrow what the driver reads: drop extras from the end
        keys = list(flat)
        while len(line) >= LINE_LIMIT and keys:
            flat.pop(keys.pop())
            line = json.dumps(out, separators=(",", ":"), allow_nan=False)
    return line


def write_detail(full: dict):
    """everything the line leaves out: bench_detail.json beside bench.py (and under gpurun_out/ so it travels back
    from a GPU box), and stderr"""
    text = json.dumps(full, indent=1)
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(text + "\n")
        except OSError:
            pass
    sys.stderr.write("bench detail: " + json.dumps(full) + "\n")
    sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=float, default=100e6, help="total rows over all ranks")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--cpu-rows", type=float, default=2e6)
    ap.add_argument("--cpu-queries", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the feedback-loop and CLIP sections")
    ap.add_argument("--loop-images", type=int, default=120000, help="images of the full-size feedback-loop dataset")
    ap.add_argument("--clustered-graph", action="store_true",
                    help="also time the label-propagation sweep on a clustered graph with the locality order (20 s of set-up)")
    args = ap.parse_args()

    # The job's stdout is ONE JSON line.  Libraries underneath write to file descriptor 1 whenever they like (RCCL's
    # version banner at communicator creation, gloo's connection notes): for the whole run descriptor 1 points at
    # stderr, and the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    # Rehearsal of the N > 1 path on a one-GPU box (SSW_BENCH_REHEARSAL=gloo): every rank uses GPU 0 and the
    # exchange goes through host tensors over gloo -- the same shards, kernels, messages and merge as the real run,
    # only the collective's transport differs (RCCL cannot put two ranks on one device).  Its numbers mean nothing.
    rehearsal = os.environ.get("SSW_BENCH_REHEARSAL") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from seesaw_amd.sharded import ShardedSyntheticIndex

    n_total = int(args.rows)
    k = args.k
    dev = torch.device("cuda", local_rank)
    t_setup = time.perf_counter()
    index = ShardedSyntheticIndex(n_total, 512, args.seed, rank, world, local_rank, k_max=max(128, k),
                                  comm_device="cpu" if (rehearsal and world > 1) else None)
    torch.cuda.synchronize(dev)
    t_index = time.perf_counter() - t_setup
    nq = args.steps + args.warmup
    q_host = np.stack([synth_query(i) for i in range(nq)])
    q_dev = torch.from_numpy(q_host).to(dev)
    torch.cuda.synchronize(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    def step(i):
        return index.topk_async(q_dev[i].data_ptr(), k)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    barrier()

    index.local.profile(True)
    if world > 1:
        index.xchg.time_collective(True)  # two event records per step around the all-gather, on its stream
    torch.cuda.synchronize(dev)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, nq):
        keys, count = step(i)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms = index.local.profile_read()
    index.local.profile(False)
    coll_us = index.xchg.collective_us() if world > 1 else []
    index.xchg.time_collective(False)
    # every message carried its shard's select-overflow flag: a set flag means some query's keys were not exact
    index.xchg.assert_no_overflow_seen()

    # last result -> host (outside the timed region) as a sanity anchor
    from seesaw_amd.device_index import decode_keys
    c = int(count.item())
    imgs, scores = decode_keys(keys[:c].cpu().numpy().view(np.uint64))

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # N > 1: the feedback loop does not shard (its scans are sub-millisecond at LVIS scale); every GPU runs
    # its own replica of the benchmark sessions, as seesaw_bench.parallel_run spreads sessions over actors.
    # Every rank takes part in the gather even if its replica failed.
    replicas = None
    if world > 1 and not args.no_extras:
        index.close()  # free the shard before the loop datasets move in
        import contextlib
        import io
        try:
            with contextlib.redirect_stdout(io.StringIO()):  # rank 0's JSON line is the only stdout of the job
                mine = feedback_loop_extras(local_rank, args.loop_images, with_cpu=False)
        except Exception as e:
            mine = {"error": f"{type(e).__name__}: {e}"}
        replicas = [None] * world
        dist.all_gather_object(replicas, mine)

    if rank == 0:
        n_local = index.n_local
        avg_ms = float(np.mean(scan_ms)) if len(scan_ms) else float("nan")
        achieved = n_local * ROW_BYTES / (avg_ms * 1e-3) / 1e9
        traffic, traffic_note = measured_traffic(n_local)
        out = {
            "metric": "vectors scanned/sec (100M×512 top-k)",
            "value": n_total * args.steps / elapsed,
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_total}x512 f32 unit-norm rows, brute-force cosine scan + exact top-{k}, "
                            f"row-sharded over {world} GPU(s) ({n_local} rows = {n_local * ROW_BYTES / 1e9:.1f} GB on rank 0), "
                            "RCCL all-gather of per-shard top-k + merge" if world > 1 else
                            f"{n_total}x512 f32 unit-norm rows ({n_total * ROW_BYTES / 1e9:.1f} GB resident), "
                            f"brute-force cosine scan + exact top-{k}, 1 GPU",
                "rows_total": n_total, "dim": 512, "k": k, "rows_per_gpu": n_local,
                "parallelism": f"row-shard x{world}",
                # the ranks the collective's backend saw (dist.get_world_size() after init_process_group) and which backend
                "rccl_ranks": (int(dist.get_world_size()) if (dist is not None and not rehearsal) else (1 if dist is None else 0)),
                "collective": ("none (one GPU)" if dist is None else
                               ("gloo over host tensors (REHEARSAL on one GPU: numbers mean nothing)" if rehearsal else
                                ("ssw_topk_allgather (ncclAllGather through the C-ABI)" if os.environ.get("SSW_C_COMM") else
                                 f"torch.distributed {dist.get_backend()} all_gather_into_tensor (RCCL), {int(dist.get_world_size())} ranks"))),
            },
            # per-step all-gather of the k-key messages, HIP events around the collective on rank 0's stream (N > 1)
            "allgather_us": ({"mean": float(np.mean(coll_us)), "median": float(np.median(coll_us)), "max": float(np.max(coll_us)),
                              "steps": len(coll_us), "bytes_per_rank": int(index.xchg.msg_len * 8)} if coll_us else None),
            "roofline": {
                "bound": "hbm", "kernel": "scan_scores_kernel<2,2,nt>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                "avg_launch_ms": avg_ms, "launches": int(len(scan_ms)),
                "algorithmic_bytes_per_launch": n_local * ROW_BYTES,
            },
            "top1": {"image": int(imgs[0]) if c else None, "score": float(scores[0]) if c else None},
        }
        t_cpu = time.perf_counter()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(index.local, k, int(args.cpu_rows), args.cpu_queries)
        else:
            out["cpu_baseline"] = None
        out["setup_seconds"] = {"index_fill_on_device": round(t_index, 2), "cpu_baseline_incl_download": round(time.perf_counter() - t_cpu, 2)}
        index.close()
        if world == 1 and not args.no_extras:
            extras = {}
            section_s = {}
            for key, fn in (("c2_one_million_rows", lambda: c2_extras(local_rank)),
                            ("sharded_step_12p5M_rows", lambda: sharded_step_extras(local_rank, k)),
                            ("feedback_fit", lambda: fit_extras(local_rank)),
                            ("feedback_loop", lambda: feedback_loop_extras(local_rank, args.loop_images, clustered_graph=args.clustered_graph)),
                            ("clip", lambda: clip_extras(local_rank))):
                t_sec = time.perf_counter()
                try:
                    extras[key] = fn()
                except Exception as e:  # the headline metric above stands on its own
                    extras[key] = {"error": f"{type(e).__name__}: {e}"}
                section_s[key] = round(time.perf_counter() - t_sec, 2)
            extras["section_seconds"] = section_s  # host wall time of each extras section, set-up included
            out["extras"] = extras
        if replicas is not None:
            out["extras"] = {"feedback_loop_replicas": aggregate_replicas(replicas, world)}
        sys.stdout.flush()
        write_detail(out)
        os.write(json_fd, (compact_line(out) + "\n").encode())
    os.close(json_fd)
    index.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
