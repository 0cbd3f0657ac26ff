"""CPU, world_size 2 over gloo: the row-sharded top-k exchange (partition, global id
offsetting, all_gather, merge) -- the N > 1 path of bench.py -- with each rank's local scan
replaced by the CPU oracle (the HIP kernels need a GPU; the collective logic does not)."""
import os

import numpy as np
import pytest

from conftest import free_port  # noqa: E402
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _encode(scores, ids):
    """composite keys exactly as the select kernels emit them."""
    u = np.ascontiguousarray(scores, dtype=np.float32).view(np.uint32).astype(np.uint64)
    neg = (u & np.uint64(0x80000000)) != 0
    o = np.where(neg, ~u & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    return (o << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.asarray(ids, dtype=np.uint64))


def _merge_on_cpu(device, stream, keys, counts, k, out_keys, out_count):
    """stand-in for ssw_topk_merge_dev: same contract (descending u64 order)."""
    lists = [keys[r, : int(counts[r])].numpy().view(np.uint64) for r in range(keys.shape[0])]
    allk = np.sort(np.concatenate(lists))[::-1][:k]
    out_keys[: allk.shape[0]] = torch.from_numpy(allk.view(np.int64).copy())
    out_count[0] = allk.shape[0]


def _worker(rank, world, port, n_total, k, tmpdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import seesaw_oracle as orc
    from seesaw_amd.device_index import decode_keys
    from seesaw_amd.sharded import ShardedTopK, shard_bounds
    lo, hi = shard_bounds(n_total, world, rank)
    X = orc.synth_rows(9, lo, hi - lo, 512)          # this rank's slice of the global index
    q = orc.synth_query(1)
    local_scores = orc.scores_kernel_order(X, q)
    ids, sc, _ = orc.topk_images_tiebreak(local_scores, None, hi - lo, [], k)
    keys = torch.zeros(4096, dtype=torch.int64)
    keys[: ids.shape[0]] = torch.from_numpy(_encode(sc, ids).view(np.int64).copy())
    count = torch.tensor([ids.shape[0]], dtype=torch.int32)
    x = ShardedTopK(rank=rank, world=world, device=torch.device("cpu"), image_offset=lo, k_max=128, merge=_merge_on_cpu)
    out_keys, out_count = x.exchange(keys, count, k)
    c = int(out_count.item())
    imgs, scores = decode_keys(out_keys[:c].numpy().view(np.uint64))
    clean = x.overflowed()
    # the select-overflow flag rides in the same message: rank 1 raises it, every rank learns of it
    flagged = torch.tensor([ids.shape[0], 1 if rank == 1 else 0], dtype=torch.int32)
    x.exchange(keys, flagged, k)
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), imgs=imgs, scores=scores, clean=np.asarray(clean, dtype=np.int64),
             flagged=np.asarray(x.overflowed(), dtype=np.int64), seen=int(x.flags_seen.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,k", [(5000, 50), (301, 100)])
def test_two_rank_exchange_matches_single_index(tmp_path, oracle, n_total, k):
    port = free_port()
    mp.spawn(_worker, args=(2, port, n_total, k, str(tmp_path)), nprocs=2, join=True)
    X = oracle.synth_rows(9, 0, n_total, 512)
    q = oracle.synth_query(1)
    ref_ids, ref_sc, _ = oracle.topk_images_tiebreak(oracle.scores_kernel_order(X, q), None, n_total, [], k)
    for r in range(2):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(g["imgs"], ref_ids), r            # every rank holds the global answer
        assert np.array_equal(g["scores"].view(np.uint32), ref_sc.view(np.uint32))
        assert g["clean"].size == 0 and g["flagged"].tolist() == [1] and int(g["seen"]) == 1


RAGGED_8 = [700, 1, 0, 2500, 13, 900, 300, 586]  # rows per rank: ragged, one single-row shard, one EMPTY shard


def _worker8(rank, world, port, sizes, k, tmpdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import seesaw_oracle as orc
    from seesaw_amd.device_index import decode_keys
    from seesaw_amd.sharded import ShardedTopK
    lo = int(np.sum(sizes[:rank]))
    n = int(sizes[rank])
    q = orc.synth_query(1)
    keys = torch.zeros(4096, dtype=torch.int64)
    n_local = 0
    if n > 0:
        X = orc.synth_rows(9, lo, n, 512)
        ids, sc, _ = orc.topk_images_tiebreak(orc.scores_kernel_order(X, q), None, n, [], min(k, n))
        keys[: ids.shape[0]] = torch.from_numpy(_encode(sc, ids).view(np.int64).copy())
        n_local = ids.shape[0]
    x = ShardedTopK(rank=rank, world=world, device=torch.device("cpu"), image_offset=lo, k_max=128, merge=_merge_on_cpu)

    def run(flag):
        if n == 0:
            x.pack_empty()
        else:
            x.pack(keys, torch.tensor([n_local, flag], dtype=torch.int32), min(k, n))
        x.gather()
        return x.merge_gathered(k)

    out_keys, out_count = run(0)
    c = int(out_count.item())
    imgs, scores = decode_keys(out_keys[:c].numpy().view(np.uint64))
    clean, seen0 = x.overflowed(), int(x.flags_seen.item())
    # the overflow flag of ONE shard (rank 5) reaches every rank in the same message; the protocol then re-exchanges
    run(1 if rank == 5 else 0)
    flagged, seen1 = x.overflowed(), int(x.flags_seen.item())
    out_keys2, out_count2 = run(0)          # rank 5 has repaired its selection: the flags are gone, the keys unchanged
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), imgs=imgs, scores=scores, clean=np.asarray(clean, dtype=np.int64),
             flagged=np.asarray(flagged, dtype=np.int64), seen=np.asarray([seen0, seen1]),
             after=np.asarray(x.overflowed(), dtype=np.int64),
             same=np.asarray(torch.equal(out_keys2[:c], out_keys[:c]) and int(out_count2.item()) == c))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [100, 128])
def test_eight_rank_exchange_ragged_empty_and_overflow_on_rank_5(tmp_path, oracle, k):
    """VERDICT r3 #5a: BASELINE configs C4 / C5 are 8-rank configurations.  Eight gloo ranks with ragged shards (2500 rows
    down to 1), one empty shard and an overflow flag raised by rank 5 only: every rank ends with the whole index's top-k
    (ids and score bits), every rank sees the flag and which rank raised it."""
    sizes = RAGGED_8
    port = free_port()
    mp.spawn(_worker8, args=(8, port, sizes, k, str(tmp_path)), nprocs=8, join=True)
    n_total = int(np.sum(sizes))
    X = oracle.synth_rows(9, 0, n_total, 512)
    q = oracle.synth_query(1)
    ref_ids, ref_sc, _ = oracle.topk_images_tiebreak(oracle.scores_kernel_order(X, q), None, n_total, [], k)
    for r in range(8):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(g["imgs"], ref_ids), r
        assert np.array_equal(g["scores"].view(np.uint32), ref_sc.view(np.uint32)), r
        assert g["clean"].size == 0 and g["flagged"].tolist() == [5] and g["seen"].tolist() == [0, 1]
        assert g["after"].size == 0 and bool(g["same"])


def test_bench_replica_aggregation():
    """bench.py at N > 1: per-GPU feedback-loop rates are summed; a failed replica is reported, not fatal"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rep = {"full": {"vectors": 10, "knn_graph": {"k": 10}, "plain": {"hip_iters_per_s": 500.0, "hip_ms_per_iter": 2.0},
                    "multi_reg": {"hip_iters_per_s": 250.0}}}
    out = bench.aggregate_replicas([rep, rep, {"error": "boom"}, None], 4)
    assert out["gpus"] == 4 and out["errors"] == ["boom"]
    assert out["aggregate"]["full"]["plain"]["iters_per_s_all_gpus"] == 1000.0
    assert out["aggregate"]["full"]["multi_reg"]["per_gpu"] == [250.0, 250.0]
