"""CPU, world_size 2 over gloo: ShardedMultiscaleIndex behind the AccessMethod interface returns, on every rank,
what the REFERENCE's unsharded MultiscaleIndex returned (tests/golden/multiscale_query.npz): four stateful rounds
with growing exclusions (plain_score), the vector2 form, and agg_method='avg_score' with all three aug_larger modes
on the 3-level tile pyramid.  The per-rank scan/select/aggregate kernels are replaced by the CPU oracle
(tests/_oracle_shard.py); partitioning, restricted exclusion, the exchange, the merge bookkeeping and the owner-side
second stage are the product's."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import free_port  # noqa: E402
import torch.distributed as dist
import torch.multiprocessing as mp

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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _oracle_shard import OracleShard, merge_on_cpu
    from oracle import seesaw_oracle as orc
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.sharded_index import ShardedMultiscaleIndex
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    out = {}
    # ---- the stateful plain_score rounds + vector2 ------------------------------------------------------
    meta, seed = _meta(g["meta"]), int(g["seed"])
    X = orc.synth_rows(seed, 0, meta.shape[0], 512)
    q = orc.synth_query(seed)
    index = ShardedMultiscaleIndex(embedding=None, vectors=X, vector_meta=meta, rank=rank, world=world,
                                   shard_factory=OracleShard, merge=merge_on_cpu, k_max=128)
    assert index.row_hi - index.row_lo < X.shape[0] and len(index) == int(g["n_images"])
    out["rows"] = np.asarray([index.row_lo, index.row_hi, index.img_lo, index.img_hi])
    qq = index.new_query()
    for rnd in range(4):
        res = qq.query_stateful(vector=q, batch_size=5, shortlist_size=50, force_exact=True, agg_method="plain_score",
                                aug_larger="all", rescore_method=None)
        out[f"r{rnd}_dbidxs"], out[f"r{rnd}_acts"] = np.asarray(res["dbidxs"]), _acts(res)
    res = index.query(vector=q, vector2=orc.synth_query(seed + 1), topk=5, shortlist_size=50, exclude=BitMap(),
                      force_exact=True, agg_method="plain_score", aug_larger="all", rescore_method=None)
    out["v2_dbidxs"], out["v2_scores"] = np.asarray(res["dbidxs"]), _acts(res)[:, 5]
    # ---- avg_score on the tile pyramid ---------------------------------------------------------------------
    pmeta, pseed = _meta(g["pyr_meta"]), int(g["pyr_seed"])
    PX = orc.synth_rows(pseed, 0, pmeta.shape[0], 512)
    pq = orc.synth_query(pseed)
    lo, hi = ShardedMultiscaleIndex.row_range(pmeta, world, rank)
    # this one is built from the rank's own rows only (no rank ever holds the whole matrix)
    pindex = ShardedMultiscaleIndex(embedding=None, vectors=None, local_vectors=PX[lo:hi], vector_meta=pmeta, rank=rank,
                                    world=world, shard_factory=OracleShard, merge=merge_on_cpu, k_max=128)
    for aug in ("all", "greater", "adjacent"):
        res = pindex.query(vector=pq, topk=10, shortlist_size=50, exclude=BitMap(pmeta.dbidx.values[:40]),
                           force_exact=True, agg_method="avg_score", aug_larger=aug, rescore_method=None)
        out[f"avg_{aug}_dbidxs"], out[f"avg_{aug}_acts"] = np.asarray(res["dbidxs"]), _acts(res)
    res = pindex.query(vector=pq, vector2=orc.synth_query(pseed + 1), topk=10, shortlist_size=50, exclude=BitMap(),
                       force_exact=True, agg_method="avg_score", aug_larger="greater", rescore_method=None)
    out["avg_v2_dbidxs"], out["avg_v2_acts"] = np.asarray(res["dbidxs"]), _acts(res)
    # round 3: rows served from the owning shards (no rank holds the matrix), ranking by caller-supplied scores
    cut = ShardedMultiscaleIndex.row_range(pmeta, world, 0)[1]  # the same rows on every rank: gather_rows is a collective
    rows = np.array([0, 5, cut - 1, cut, PX.shape[0] - 1, 17, 17])
    out["rows_ok"] = np.asarray(np.array_equal(pindex.vectors[rows], PX[rows]) and pindex.vectors.shape == PX.shape
                                and np.array_equal(np.asarray(pindex.vectors), PX))
    fake = np.sin(np.arange(PX.shape[0], dtype=np.float64) * 0.37)  # any per-row score every rank agrees on
    skip = np.zeros(PX.shape[0], dtype=bool)
    skip[::7] = True
    cand = pindex.topk_from_scores(fake, topk_dbidx=20, exclude_dbidx=BitMap(pmeta.dbidx.values[:40]), skip_rows=skip)
    out["tfs_pos"], out["tfs_scores"], out["tfs_rows"] = cand.attrs["positions"], np.asarray(cand.max_score), cand.attrs["best_rows"]
    # the full score vector, assembled from the slices
    out["score_head"] = pindex.score(pq)[:64]
    out["score_len"] = np.asarray(pindex.score(pq).shape[0])
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_index_matches_reference_on_every_rank(tmp_path, oracle, world):
    """2 ranks, and the 8 of BASELINE configs C4 / C5 (VERDICT r3 #5a): every rank returns the reference's own answers"""
    port = free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, "multiscale_query.npz"))
    seed = int(g["seed"])
    X = oracle.synth_rows(seed, 0, g["meta"].shape[0], 512)
    band = oracle.rounding_band(X, oracle.synth_query(seed))
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    for a, b in zip(r[:-1], r[1:]):
        assert a["rows"][1] == b["rows"][0] and a["rows"][3] == b["rows"][2]   # contiguous, disjoint
    assert r[0]["rows"][0] == 0 and r[-1]["rows"][1] == g["meta"].shape[0]
    for k in range(world):
        for rnd in range(4):
            assert np.array_equal(r[k][f"r{rnd}_dbidxs"], g[f"r{rnd}_dbidxs"]), (k, rnd)
            ref = g[f"r{rnd}_activations"]
            assert np.array_equal(r[k][f"r{rnd}_acts"][:, :5], ref[:, :5])
            assert np.abs(r[k][f"r{rnd}_acts"][:, 5] - ref[:, 5]).max() <= band
        assert np.array_equal(r[k]["v2_dbidxs"], g["v2_dbidxs"])
        assert np.abs(r[k]["v2_scores"] - g["v2_scores"]).max() <= 2 * band
        for tag in ("avg_all", "avg_greater", "avg_adjacent", "avg_v2"):
            assert np.array_equal(r[k][f"{tag}_dbidxs"], g[f"{tag}_dbidxs"]), (k, tag)
            ref = g[f"{tag}_activations"]
            assert np.array_equal(r[k][f"{tag}_acts"][:, :5], ref[:, :5]), (k, tag)
            assert np.abs(r[k][f"{tag}_acts"][:, 5] - ref[:, 5]).max() <= 1e-6
        assert int(r[k]["score_len"]) == g["pyr_meta"].shape[0]
        assert bool(r[k]["rows_ok"])
    for k in range(1, world):
        assert np.array_equal(r[0]["score_head"], r[k]["score_head"])
    # topk_from_scores: both ranks agree, and agree with the selection on the unsharded arrays
    pm = g["pyr_meta"]
    fake = np.sin(np.arange(pm.shape[0], dtype=np.float64) * 0.37).astype(np.float32)
    fake[::7] = -np.inf
    dbidx, r2i = np.unique(pm[:, 0].astype(np.int64), return_inverse=True)
    excl = np.searchsorted(dbidx, np.unique(pm[:40, 0].astype(np.int64)))
    ids, sc, rows = oracle.topk_images_tiebreak(fake, r2i, dbidx.shape[0], list(excl), 20)
    for k in range(world):
        assert np.array_equal(r[k]["tfs_pos"], ids) and np.array_equal(r[k]["tfs_rows"], rows)
        assert np.array_equal(r[k]["tfs_scores"].astype(np.float32), sc)


def test_shard_bounds_by_image_cover_and_balance():
    from seesaw_amd.sharded import shard_bounds_by_image
    rng = np.random.default_rng(0)
    tiles = rng.integers(1, 40, 1000)
    row_start = np.concatenate(([0], np.cumsum(tiles)))
    for world in (1, 2, 3, 8):
        cuts = [shard_bounds_by_image(row_start, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 1000 and cuts[-1][3] == row_start[-1]
        for a, b in zip(cuts[:-1], cuts[1:]):
            assert a[1] == b[0] and a[3] == b[2]
        sizes = [c[3] - c[2] for c in cuts]
        assert max(sizes) - min(sizes) <= 2 * 40


# ---------------------------------------------------------------------------------------------------------------------
# world 8 with an EMPTY shard, an overflow flag on rank 5, and a whole session (VERDICT r3 #5a)
# ---------------------------------------------------------------------------------------------------------------------
def _skewed_meta(seed=3):
    """an index whose row-balanced cut leaves one of eight shards without an image: image 40 alone holds a quarter of
    the rows (a panorama with 1 300 tiles), so two consecutive cuts fall inside it"""
    rng = np.random.default_rng(seed)
    counts = rng.integers(1, 30, size=260)
    counts[40] = 1300
    dbidx = np.repeat(np.arange(260) * 3 + 1, counts)  # dbidx values are not positions
    n = dbidx.shape[0]
    return pd.DataFrame({"dbidx": dbidx.astype(np.int64), "zoom_level": np.zeros(n, np.int16),
                         "x1": np.zeros(n, np.float32), "y1": np.zeros(n, np.float32),
                         "x2": np.full(n, 224, np.float32), "y2": np.full(n, 224, np.float32)})


def _worker8_special(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    import io
    import json
    import torch
    from _oracle_shard import OracleShard, merge_on_cpu
    from oracle import seesaw_oracle as orc
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.indices.multiscale.sharded_index import ShardedMultiscaleIndex
    out = {}
    meta = _skewed_meta()
    X = orc.synth_rows(77, 0, meta.shape[0], 512)
    calls = {"fast": 0, "deep": 0}

    class FlaggingShard(OracleShard):
        """rank 5's fast selection reports an overflow (as the HIP selection does on mass ties) the first time of every
        query; its deep selection is the exact one"""

        def select(self, q, k, excluded_local):
            keys, count, best = super().select(q, k, excluded_local)
            calls["fast"] += 1
            self._last = (q, k, excluded_local)
            if rank == 5:
                keys = keys.clone()
                keys[: int(count[0])] = 0          # what an overflowed selection leaves is not to be trusted
                count = torch.tensor([int(count[0]), 1], dtype=torch.int32)
            return keys, count, best

        def select_deep(self, k):
            calls["deep"] += 1
            q, _, excluded_local = self._last
            return OracleShard.select(self, q, k, excluded_local)

    index = ShardedMultiscaleIndex(embedding=None, vectors=X, vector_meta=meta, rank=rank, world=world,
                                   shard_factory=FlaggingShard, merge=merge_on_cpu, k_max=128)
    out["bounds"] = np.asarray([index.img_lo, index.img_hi, index.row_lo, index.row_hi])
    qq = index.new_query()
    for rnd in range(3):
        res = qq.query_stateful(vector=orc.synth_query(5 + rnd), batch_size=7, shortlist_size=60, force_exact=True,
                                agg_method="plain_score", aug_larger="all", rescore_method=None)
        out[f"r{rnd}_dbidxs"] = np.asarray(res["dbidxs"])
        out[f"r{rnd}_scores"] = np.asarray([float(a.score.values[0]) for a in res["activations"]])
    out["calls"] = np.asarray([calls["fast"], calls["deep"]])
    index.close()
    # ---- a whole `plain` session of the reference's bench fixture over eight shards
    from seesaw_amd.basic_types import BenchParams, IndexSpec, SessionParams
    from seesaw_amd.seesaw_bench import benchmark_loop
    from seesaw_amd.seesaw_session import Session
    from seesaw_amd.synthetic import GlobalDataManager, make_dataset
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    spec = json.loads(str(g["datasets"]))["A"]
    ds = make_dataset("lvis", knn_k=10, **spec["make"])
    ds.embedding.noise = spec["noise"]
    gdm = GlobalDataManager().add(ds)
    boxes, _ = ds.load_ground_truth()
    lo, hi = ShardedMultiscaleIndex.row_range(ds.vector_meta, world, rank)
    sidx = ShardedMultiscaleIndex(embedding=ds.embedding, vectors=None, local_vectors=ds.vectors[lo:hi], vector_meta=ds.vector_meta,
                                  rank=rank, world=world, shard_factory=OracleShard, merge=merge_on_cpu, k_max=128)
    p = SessionParams(index_spec=IndexSpec(d_name="lvis", i_name="multiscale", c_name=None), interactive="plain",
                      interactive_options=None, shortlist_size=50, agg_method="plain_score", aug_larger="greater", batch_size=1,
                      start_policy="after_first_batch", index_options={"use_vec_index": False})
    b = BenchParams(name="plain", ground_truth_category="c1", qstr="a c1", n_batches=25, max_results=10)
    np.random.seed(0)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        session = Session(gdm, ds, sidx, p)
        res = benchmark_loop(session=session, box_data=boxes, subset=BitMap(ds.file_meta.index.values), b=b, p=p)
    out["plain_shown"] = np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1) for a in session.acc_indices])
    out["plain_nfound"] = np.asarray(res["nfound"])
    sidx.close()
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_empty_shard_overflow_on_rank_5_and_a_session(tmp_path, oracle):
    port = free_port()
    mp.spawn(_worker8_special, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(8)]
    n_imgs = [int(x["bounds"][1] - x["bounds"][0]) for x in r]
    assert 0 in n_imgs and sum(n_imgs) == 260, n_imgs                     # one shard holds no image
    assert r[5]["bounds"][1] > r[5]["bounds"][0]                          # the flagging rank is not the empty one
    # the unsharded answer: the oracle's selection over the whole index, round after round with the exclusions growing
    meta = _skewed_meta()
    X = oracle.synth_rows(77, 0, meta.shape[0], 512)
    dbidx, r2i = np.unique(meta.dbidx.values, return_inverse=True)
    shown = []
    for rnd in range(3):
        sc = oracle.scores_kernel_order(X, oracle.synth_query(5 + rnd))
        excl = np.searchsorted(dbidx, np.asarray(shown, dtype=np.int64))
        ids, best, _ = oracle.topk_images_tiebreak(sc, r2i, dbidx.shape[0], list(excl), 7)
        for k in range(8):
            assert np.array_equal(r[k][f"r{rnd}_dbidxs"], dbidx[ids]), (k, rnd)
            assert np.array_equal(r[k][f"r{rnd}_scores"].astype(np.float32).view(np.uint32), best.view(np.uint32)), (k, rnd)
        shown.extend(dbidx[ids].tolist())
    for k in range(8):
        fast, deep = r[k]["calls"].tolist()
        assert fast == (3 if n_imgs[k] else 0) and deep == (3 if k == 5 else 0), (k, fast, deep)  # only rank 5 repairs
    g = np.load(os.path.join(GOLDEN, "bench_loop.npz"))
    for k in range(8):                                                     # the reference's own session, on every rank
        assert np.array_equal(r[k]["plain_shown"], g["plain_shown"]), k
        assert int(r[k]["plain_nfound"]) == int(g["plain_nfound"])
