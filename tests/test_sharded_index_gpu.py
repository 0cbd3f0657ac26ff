"""GPU: ShardedMultiscaleIndex with the real kernels.  Two ranks share the box's one GPU (each holds its own slice
in HBM and runs the HIP scan / select / merge / avg_score kernels on it); the process group is gloo with the
messages staged through the host (`comm_device="cpu"`) because RCCL refuses two ranks on one device -- the RCCL
collective itself is covered by tests/test_sharded_gpu.py::test_exchange_over_rccl_world_size_1.  Every rank must
return what the REFERENCE's unsharded MultiscaleIndex returned (tests/golden/multiscale_query.npz)."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import free_port  # noqa: E402
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _meta(m):
    return pd.DataFrame({"dbidx": m[:, 0].astype(np.int64), "zoom_level": m[:, 1].astype(np.int16),
                         "x1": m[:, 2].astype(np.float32), "y1": m[:, 3].astype(np.float32),
                         "x2": m[:, 4].astype(np.float32), "y2": m[:, 5].astype(np.float32)})


def _acts(res):
    return np.stack([a[["x1", "y1", "x2", "y2", "dbidx", "score"]].values[0].astype(np.float64) for a in res["activations"]])


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    torch.cuda.set_device(0)
    from oracle import seesaw_oracle as orc
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.sharded_index import ShardedMultiscaleIndex
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    out = {}
    meta, seed = _meta(g["meta"]), int(g["seed"])
    X = orc.synth_rows(seed, 0, meta.shape[0], 512)
    q = orc.synth_query(seed)
    lo, hi = ShardedMultiscaleIndex.row_range(meta, world, rank)
    index = ShardedMultiscaleIndex(embedding=None, vectors=None, local_vectors=X[lo:hi], vector_meta=meta, rank=rank,
                                   world=world, device=0, comm_device="cpu", k_max=128)
    qq = index.new_query()
    for rnd in range(4):
        res = qq.query_stateful(vector=q, batch_size=5, shortlist_size=50, force_exact=True, agg_method="plain_score",
                                aug_larger="all", rescore_method=None)
        out[f"r{rnd}_dbidxs"], out[f"r{rnd}_acts"] = np.asarray(res["dbidxs"]), _acts(res)
    res = index.query(vector=q, vector2=orc.synth_query(seed + 1), topk=5, shortlist_size=50, exclude=BitMap(),
                      force_exact=True, agg_method="plain_score", aug_larger="all", rescore_method=None)
    out["v2_dbidxs"], out["v2_scores"] = np.asarray(res["dbidxs"]), _acts(res)[:, 5]
    pmeta, pseed = _meta(g["pyr_meta"]), int(g["pyr_seed"])
    PX = orc.synth_rows(pseed, 0, pmeta.shape[0], 512)
    pq = orc.synth_query(pseed)
    pindex = ShardedMultiscaleIndex(embedding=None, vectors=PX, vector_meta=pmeta, rank=rank, world=world, device=0,
                                    comm_device="cpu", k_max=128)
    for aug in ("all", "greater", "adjacent"):
        res = pindex.query(vector=pq, topk=10, shortlist_size=50, exclude=BitMap(pmeta.dbidx.values[:40]),
                           force_exact=True, agg_method="avg_score", aug_larger=aug, rescore_method=None)
        out[f"avg_{aug}_dbidxs"], out[f"avg_{aug}_acts"] = np.asarray(res["dbidxs"]), _acts(res)
    res = pindex.query(vector=pq, vector2=orc.synth_query(pseed + 1), topk=10, shortlist_size=50, exclude=BitMap(),
                       force_exact=True, agg_method="avg_score", aug_larger="greater", rescore_method=None)
    out["avg_v2_dbidxs"], out["avg_v2_acts"] = np.asarray(res["dbidxs"]), _acts(res)
    full = pindex.score(pq)
    out["score_ok"] = np.asarray(np.array_equal(full.view(np.uint32), orc.scores_kernel_order(PX, pq).view(np.uint32)))
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), **out)
    index.close()
    pindex.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_match_reference(tmp_path, oracle):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    seed = int(g["seed"])
    band = oracle.rounding_band(oracle.synth_rows(seed, 0, g["meta"].shape[0], 512), oracle.synth_query(seed))
    for k in range(2):
        r = np.load(tmp_path / f"rank{k}.npz")
        for rnd in range(4):
            assert np.array_equal(r[f"r{rnd}_dbidxs"], g[f"r{rnd}_dbidxs"]), (k, rnd)
            ref = g[f"r{rnd}_activations"]
            assert np.array_equal(r[f"r{rnd}_acts"][:, :5], ref[:, :5])
            assert np.abs(r[f"r{rnd}_acts"][:, 5] - ref[:, 5]).max() <= band
        assert np.array_equal(r["v2_dbidxs"], g["v2_dbidxs"])
        assert np.abs(r["v2_scores"] - g["v2_scores"]).max() <= 2 * band
        for tag in ("avg_all", "avg_greater", "avg_adjacent", "avg_v2"):
            assert np.array_equal(r[f"{tag}_dbidxs"], g[f"{tag}_dbidxs"]), (k, tag)
            ref = g[f"{tag}_activations"]
            assert np.array_equal(r[f"{tag}_acts"][:, :5], ref[:, :5]), (k, tag)
            assert np.abs(r[f"{tag}_acts"][:, 5] - ref[:, 5]).max() <= 1e-6
        assert bool(r["score_ok"])


def test_world_size_1_equals_unsharded(oracle):
    """the degenerate shard: identical to MultiscaleIndex, including a session's stateful rounds"""
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex
    from seesaw_amd.indices.multiscale.sharded_index import ShardedMultiscaleIndex
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    meta, seed = _meta(g["pyr_meta"]), int(g["pyr_seed"])
    X = oracle.synth_rows(seed, 0, meta.shape[0], 512)
    q = oracle.synth_query(seed)
    a = MultiscaleIndex(embedding=None, vectors=X, vector_meta=meta)
    b = ShardedMultiscaleIndex(embedding=None, vectors=X, vector_meta=meta, rank=0, world=1, k_max=256)
    qa, qb = a.new_query(), b.new_query()
    for agg, aug in (("plain_score", "all"), ("avg_score", "greater"), ("avg_score", "all")):
        for _ in range(3):
            kw = dict(vector=q, batch_size=7, shortlist_size=60, force_exact=True, agg_method=agg, aug_larger=aug,
                      rescore_method=None)
            ra, rb = qa.query_stateful(**kw), qb.query_stateful(**kw)
            assert np.array_equal(ra["dbidxs"], rb["dbidxs"])
            assert np.array_equal(_acts(ra), _acts(rb))
    b.close()


def _session_worker(rank, world, port, tmpdir):
    """whole feedback sessions over the sharded index: no rank holds the matrix (vectors=None), the fitting loops read
    the labelled rows through index.vectors[rows] = gather_rows (owning shard + all-reduce), the graph loop ranks
    through topk_from_scores on the shards"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    import io
    import json
    import torch
    torch.cuda.set_device(0)
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.sharded_index import ShardedMultiscaleIndex
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import Session
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    spec = json.loads(str(g["datasets"]))["A"]
    ds = make_dataset("lvis", knn_k=10, **spec["make"])
    ds.embedding.noise = spec["noise"]
    gdm = GlobalDataManager().add(ds)
    boxes, _ = ds.load_ground_truth()
    graph = ds.knn_graph()
    matrix = dict(knn_path="nndescent60", symmetric=True, self_edges=False, normalized_weights=False, knn_k=10, edist=0.05)
    options = {
        "plain": None,
        "multi_reg_data": dict(label_loss_type="pairwise_rank_loss", rank_loss_margin=0.2, use_qvec_norm=None,
                               reg_data_lambda=1000.0, reg_norm_lambda=100.0, reg_query_lambda=10.0, verbose=False,
                               max_iter=100, pos_weight="balanced", lr=1.0, matrix_options=matrix),
        "knn_prop2": dict(matrix_options=matrix, normalize_scores=False, sigmoid_before_propagate=True, calib_a=10.0,
                          calib_b=-0.4, prior_weight=1.0),
    }
    lo, hi = ShardedMultiscaleIndex.row_range(ds.vector_meta, world, rank)
    out = {}
    for name, opts in options.items():
        index = ShardedMultiscaleIndex(embedding=ds.embedding, vectors=None, local_vectors=ds.vectors[lo:hi],
                                       vector_meta=ds.vector_meta, rank=rank, world=world, device=0, comm_device="cpu",
                                       k_max=128)
        index.knng = {n_: graph for n_ in ("exact", "nndescent60", "")}
        # gather_rows against the host array (which this rank would not have in production)
        cut = ShardedMultiscaleIndex.row_range(ds.vector_meta, world, 0)[1]  # the same rows on every rank (a collective)
        rows = np.array([0, 7, cut - 1, cut, ds.vectors.shape[0] - 1, 7])
        assert np.array_equal(index.vectors[rows], ds.vectors[rows]) and index.vectors.shape == ds.vectors.shape
        interactive = "multi_reg" if name.startswith("multi_reg") else name
        p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale", c_name=None), interactive=interactive,
                          interactive_options=opts, shortlist_size=50, agg_method="plain_score", aug_larger="greater",
                          batch_size=1, start_policy="from_start" if name == "knn_prop2" else "after_first_batch",
                          index_options={"use_vec_index": False})
        b = BenchParams(name=name, ground_truth_category="c1", qstr="a c1", n_batches=25, max_results=10)
        np.random.seed(0)
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            session = Session(gdm, ds, index, p)
            res = benchmark_loop(session=session, box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
        out[f"{name}_shown"] = np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1) for a in session.acc_indices])
        out[f"{name}_nfound"] = np.asarray(res["nfound"])
        index.close()
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_sessions_over_the_sharded_index_match_reference(tmp_path):
    """plain / multi_reg (data + query regularisers) / knn_prop2 sessions over a two-rank sharded index built WITHOUT a
    host copy of the matrix return, on every rank, the reference's own sessions (tests/golden/bench_loop.npz)"""
    port = free_port()
    mp.spawn(_session_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    for k in range(2):
        r = np.load(tmp_path / f"rank{k}.npz")
        for name in ("plain", "multi_reg_data", "knn_prop2"):
            assert np.array_equal(r[f"{name}_shown"], g[f"{name}_shown"]), (k, name, r[f"{name}_shown"].tolist(),
                                                                            g[f"{name}_shown"].tolist())
            assert int(r[f"{name}_nfound"]) == int(g[f"{name}_nfound"])
