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
    port = 29900 + os.getpid() % 1000
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
