"""GPU: the code that exists only for the row-sharded index (BASELINE config C4), on ONE GPU.

* several shards of one synthetic index live side by side on the device; each runs ssw_index_topk_dev, their
  messages are packed exactly as ShardedTopK sends them, stacked, and merged by ssw_topk_merge_dev (k_final in
  list mode with n_lists > 1): the merged keys equal the whole index's keys bit for bit;
* the select-overflow flag travels with the message: > 8192 duplicated vectors split over shards are caught
  and repaired by the deep selection;
* ShardedTopK.exchange over RCCL itself: a world-size-1 `nccl` process group launched through
  torch.distributed.run runs all_gather_into_tensor on device tensors on real hardware."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _views(idx, torch):
    from seesaw_amd import _lib
    from seesaw_amd.sharded import _DevArray
    keys_ptr, count_ptr, _ = idx.result_ptrs()
    dev = torch.device("cuda", 0)
    return (torch.as_tensor(_DevArray(keys_ptr, (_lib.SSW_MAX_TOPK,), "<i8"), device=dev),
            torch.as_tensor(_DevArray(count_ptr, (2,), "<i4"), device=dev))


def _merge_shards(shards, offsets, q_dev, k, torch, k_max):
    """what `world` ranks would do, serially on one GPU: local select, pack, (stack instead of all-gather), merge"""
    from seesaw_amd.sharded import ShardedTopK
    world = len(shards)
    x = ShardedTopK(rank=0, world=world, device=torch.device("cuda", 0), image_offset=0, k_max=k_max)
    x.world = 1  # no process group: the messages are stacked by hand below
    stream = torch.cuda.current_stream().cuda_stream
    for r, (idx, off) in enumerate(zip(shards, offsets)):
        idx.set_stream(stream)
        idx.topk_dev(q_dev.data_ptr(), k)
        keys, count = _views(idx, torch)
        if idx.n_rows == 0:
            count = torch.zeros(2, dtype=torch.int32, device=keys.device)
        x.all_buf[r] = x.pack(keys, count, k, image_offset=off)
    out_keys, out_count = x.merge_gathered(k)
    torch.cuda.synchronize()
    return x, out_keys[: int(out_count.item())].cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("k,k_max", [(100, 128), (1024, 1024)])
@pytest.mark.parametrize("sizes", [[25000] * 8, [1, 40000, 0, 700, 99999, 13, 30000, 29287]])
def test_fused_message_path_equals_pack_and_merge(k, k_max, sizes):
    """round 3: the selection's last kernel writes the rank's message itself (ssw_index_set_exchange_target) and the
    merge kernel unpacks counts / overflow flags (ssw_topk_merge_msgs_dev) -- no torch elementwise kernels around the
    collective.  Message by message and merged key by key identical to pack() + merge_gathered(); best rows too."""
    import torch
    from oracle import seesaw_oracle as orc
    from seesaw_amd import _lib
    from seesaw_amd.device_index import DeviceIndex
    from seesaw_amd.sharded import ShardedTopK, _DevArray
    seed = 91
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(int)
    q_dev = torch.from_numpy(orc.synth_query(6)).cuda()
    dev = torch.device("cuda", 0)
    world = len(sizes)
    stream = torch.cuda.current_stream().cuda_stream
    ref = ShardedTopK(rank=0, world=world, device=dev, image_offset=0, k_max=k_max, with_best=True)
    fused = ShardedTopK(rank=0, world=world, device=dev, image_offset=0, k_max=k_max, with_best=True)
    ref.world = fused.world = 1
    shards = []
    for r, (n, off) in enumerate(zip(sizes, offsets)):
        idx = DeviceIndex.synthetic(n, 512, seed=seed, first_row=int(off))
        shards.append(idx)
        idx.set_stream(stream)
        one = ShardedTopK(rank=r, world=world, device=dev, image_offset=int(off), k_max=k_max, with_best=True)
        one.attach(idx, row_offset=int(off))
        idx.topk_dev(q_dev.data_ptr(), min(k, max(n, 1)) if n else k)
        keys, count = _views(idx, torch)
        _, _, best_ptr = idx.result_ptrs()
        best = torch.as_tensor(_DevArray(best_ptr, (_lib.SSW_MAX_TOPK,), "<u4"), device=dev).to(torch.int64)
        torch.cuda.synchronize()
        if n == 0:  # an empty shard launches no selection: ssw_index_topk_dev zeroes its message's count word
            assert int(one.send_buf[-1].item()) == 0
            ref.all_buf[r] = ref.pack_empty()
        else:
            kk = min(k, n)
            ref.all_buf[r] = ref.pack(keys, count, kk, image_offset=int(off), best_rows=best + int(off))
            c = int(count[0].item())
            assert torch.equal(one.send_buf[:c], ref.all_buf[r, :c])                       # keys
            assert torch.equal(one.send_buf[k_max:k_max + c], ref.all_buf[r, k_max:k_max + c])  # best rows
            assert int(one.send_buf[-1].item()) == int(ref.all_buf[r, -1].item())          # count | overflow << 32
        fused.all_buf[r] = one.send_buf
    want_keys, want_count = ref.merge_gathered(k)
    _lib.call("ssw_topk_merge_msgs_dev", 0, ctypes.c_void_p(stream), ctypes.c_void_p(fused.all_buf.data_ptr()), world, k_max, 1, k,
              ctypes.c_void_p(fused.out_keys.data_ptr()), ctypes.c_void_p(fused.out_count.data_ptr()),
              ctypes.c_void_p(fused.flags.data_ptr()), ctypes.c_void_p(fused.flags_seen.data_ptr()))
    torch.cuda.synchronize()
    c = int(want_count.item())
    assert int(fused.out_count.item()) == c == min(k, int(np.sum(sizes)))
    assert torch.equal(fused.out_keys[:c], want_keys[:c])
    assert int(fused.flags_seen.item()) == 0 and not fused.overflowed()
    for s_ in shards:
        s_.close()


@pytest.mark.parametrize("k", [100, 1024])
@pytest.mark.parametrize("sizes", [[25000] * 8, [1, 40000, 0, 700, 99999, 13, 30000, 29287]])
def test_multi_list_merge_equals_whole_index(k, sizes):
    import torch
    from oracle import seesaw_oracle as orc
    from seesaw_amd.device_index import DeviceIndex, decode_keys
    n_total, seed = int(np.sum(sizes)), 77
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(int)
    q = orc.synth_query(5)
    q_dev = torch.from_numpy(q).cuda()
    whole = DeviceIndex.synthetic(n_total, 512, seed=seed)
    whole.set_stream(torch.cuda.current_stream().cuda_stream)
    whole.topk_dev(q_dev.data_ptr(), k)
    wk, wc = _views(whole, torch)
    torch.cuda.synchronize()
    want = wk[: int(wc[0].item())].cpu().numpy().view(np.uint64).copy()
    assert want.shape[0] == min(k, n_total) and int(wc[1].item()) == 0
    shards = [DeviceIndex.synthetic(n, 512, seed=seed, first_row=int(o)) for n, o in zip(sizes, offsets)]
    # pack() subtracts the shard's first image position from the key, i.e. adds it to the image id
    x, got = _merge_shards(shards, [int(o) for o in offsets], q_dev, k, torch, k_max=1024)
    assert np.array_equal(got, want)
    assert np.all(got[:-1] > got[1:])  # strictly descending: composite keys are unique
    imgs, scores = decode_keys(got)
    assert imgs.max() < n_total and not x.overflowed()
    for s in shards + [whole]:
        s.close()


def test_pack_uses_positive_offsets_like_the_ranks_do():
    """the ranks call pack with image_offset = first image of the shard (positive); ids come out global"""
    import torch
    from oracle import seesaw_oracle as orc
    from seesaw_amd.device_index import DeviceIndex, decode_keys
    from seesaw_amd.sharded import ShardedTopK
    n, lo, k = 5000, 123456, 20
    q = orc.synth_query(2)
    q_dev = torch.from_numpy(q).cuda()
    shard = DeviceIndex.synthetic(n, 512, seed=3, first_row=lo)
    shard.set_stream(torch.cuda.current_stream().cuda_stream)
    shard.topk_dev(q_dev.data_ptr(), k)
    keys, count = _views(shard, torch)
    x = ShardedTopK(rank=0, world=1, device=torch.device("cuda", 0), image_offset=lo, k_max=128)
    out_keys, out_count = x.exchange(keys, count, k)
    torch.cuda.synchronize()
    imgs, scores = decode_keys(out_keys[: int(out_count.item())].cpu().numpy().view(np.uint64))
    local_imgs, local_scores, _ = shard.topk_fetch(k)
    assert np.array_equal(imgs, local_imgs + lo)
    assert np.array_equal(scores.view(np.uint32), local_scores.view(np.uint32))
    shard.close()


def test_overflow_flag_travels_and_deep_selection_repairs():
    """30 000 copies of one vector, split over 3 shards (2 of them beyond the 8192-candidate fast path): the
    flags arrive with the messages, the flagged shards redo their selection exactly, the merged top-k is the
    whole index's (ties: lowest image first)."""
    import torch
    from oracle import seesaw_oracle as orc
    from seesaw_amd.device_index import DeviceIndex, decode_keys
    rng = np.random.default_rng(0)
    k = 100
    q = orc.synth_query(9)
    dup = (q + 0.05 * orc.synth_query(10)).astype(np.float32)
    dup /= np.linalg.norm(dup)
    sizes = [12000, 5000, 13000]   # duplicates per shard; each shard also holds 3000 ordinary rows
    mats, offsets, off = [], [], 0
    for s in sizes:
        other = orc.synth_rows(40 + len(mats), 0, 3000, 512)
        X = np.concatenate([np.tile(dup, (s, 1)), other]).astype(np.float32)
        X = X[rng.permutation(X.shape[0])]
        mats.append(X)
        offsets.append(off)
        off += X.shape[0]
    q_dev = torch.from_numpy(q).cuda()
    shards = [DeviceIndex.from_numpy(X) for X in mats]
    x, got = _merge_shards(shards, offsets, q_dev, k, torch, k_max=128)
    flagged = x.overflowed()
    assert flagged == [0, 2], flagged                       # 12000 and 13000 > 8192; 5000 fits the fast path
    with pytest.raises(RuntimeError, match="overflowed"):
        x.assert_no_overflow_seen()
    # repair: what ShardedSyntheticIndex.topk does on every rank
    for r in flagged:
        shards[r].select_deep_dev(k)
        keys, count = _views(shards[r], torch)
        x.all_buf[r] = x.pack(keys, count, k, image_offset=offsets[r])
    out_keys, out_count = x.merge_gathered(k)
    torch.cuda.synchronize()
    assert not x.overflowed()
    got = out_keys[: int(out_count.item())].cpu().numpy().view(np.uint64)
    whole = DeviceIndex.from_numpy(np.concatenate(mats))
    imgs, scores, _ = whole.topk(q, k)                      # host path: reruns the deep selection itself
    g_imgs, g_scores = decode_keys(got)
    assert np.array_equal(g_imgs, imgs) and np.array_equal(g_scores.view(np.uint32), scores.view(np.uint32))
    allX = np.concatenate(mats)
    is_dup = np.all(allX == dup, axis=1)
    assert np.array_equal(imgs, np.nonzero(is_dup)[0][:k])   # all ties: the k lowest positions
    for s in shards + [whole]:
        s.close()


def test_exchange_over_rccl_world_size_1():
    """ShardedSyntheticIndex + ShardedTopK.exchange with backend nccl (= RCCL) initialised for real: the
    process is started through torch.distributed.run before anything touches the GPU."""
    script = os.path.join(ROOT, "tests", "rccl_world1_worker.py")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script],
                          env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((proc.stdout + proc.stderr).splitlines()[-30:])
    assert proc.returncode == 0, tail
    assert "RCCL_WORLD1_OK" in proc.stdout, tail


def test_bench_with_two_ranks_on_one_gpu_returns_the_single_gpu_answer():
    """bench.py's N > 1 path end to end (shards, selection writing the exchange message, all-gather, merge from the
    messages, overflow check, max-over-ranks timing, rank 0's JSON line) with two ranks sharing GPU 0 and the collective
    over gloo (SSW_BENCH_REHEARSAL: RCCL cannot put two ranks on one device): the last query's top-1 equals the N = 1
    run's, bit for bit."""
    import json
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    args = ["--steps", "3", "--warmup", "1", "--rows", "3000000", "--no-extras", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, env=env,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, "\n".join((one.stdout + one.stderr).splitlines()[-20:])
    env2 = dict(env)
    env2["SSW_BENCH_REHEARSAL"] = "gloo"
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env2, capture_output=True, text=True,
                         timeout=600)
    assert two.returncode == 0, "\n".join((two.stdout + two.stderr).splitlines()[-20:])
    line = lambda out: json.loads([x for x in out.splitlines() if x.startswith("{")][-1])  # noqa: E731
    a, b = line(one.stdout), line(two.stdout)
    assert b["n_gpus"] == 2 and b["config"]["rows_per_gpu"] == 1500000
    assert a["top1"] == b["top1"], (a["top1"], b["top1"])


def test_k_beyond_the_exchange_targets_k_max_is_refused_and_writes_nothing():
    """with an exchange target attached the selection's last kernel writes k words of the message; k > k_max
    used to run over the count word and past send_buf before the merge reported it.  Now the selection itself refuses
    (all three forms: one-launch small index, histogram path, deep path) and the buffers stay as they were."""
    import torch
    from oracle import seesaw_oracle as orc
    from seesaw_amd import _lib
    from seesaw_amd.device_index import DeviceIndex
    from seesaw_amd.sharded import ShardedTopK
    dev = torch.device("cuda", 0)
    q_dev = torch.from_numpy(orc.synth_query(6)).cuda()
    for n in (3000, 40000):  # one-launch form / histogram form
        idx = DeviceIndex.synthetic(n, 512, seed=3)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
        x = ShardedTopK(rank=0, world=1, device=dev, image_offset=0, k_max=16, with_best=True)
        guard = torch.full((x.msg_len + 64,), 0x5A5A5A5A, dtype=torch.int64, device=dev)
        x.send_buf = guard[:x.msg_len]  # the message sits inside a larger buffer whose tail must stay untouched
        x.attach(idx)
        idx.topk_dev(q_dev.data_ptr(), 16)
        torch.cuda.synchronize()
        before = guard.clone()
        assert int(before[x.msg_len - 1].item()) & 0xFFFFFFFF == 16
        with pytest.raises(_lib.SeesawHipError, match="k_max"):
            idx.topk_dev(q_dev.data_ptr(), 17)
        with pytest.raises(_lib.SeesawHipError, match="k_max"):
            idx.select_deep_dev(17)
        with pytest.raises(ValueError, match="k_max"):
            x.exchange_fused(17)
        torch.cuda.synchronize()
        assert torch.equal(guard, before)
        idx.close()
