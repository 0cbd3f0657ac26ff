"""Row-sharded index: one process per GPU, each scanning its contiguous slice of the rows,
one RCCL all-gather of the per-shard top-k over xGMI, deterministic merge on every rank.

The reference has no counterpart (its index is a single numpy array in host RAM,
seesaw/indices/multiscale/multiscale_index.py:220); this is the multi-GPU form of the
same `_get_top_exact` + `_get_top_dbidxs` selection (multiscale_index.py:170-199).

Shards are cut at image boundaries (rows are stored sorted by image, coarse_index.py:49),
so the per-image max never crosses a rank and the only exchange is k x 8 bytes per rank:
the composite keys (score_key << 32 | ~image) produced by the select kernels, made global
by subtracting the shard's first image position.  The payload is ~1 KB per rank, i.e. the
collective is latency-bound: one fused all_gather_into_tensor, never a ring of sends.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

from . import _lib


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of n_total units: [lo, hi) of `rank`."""
    assert 0 <= rank < world
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def shard_bounds_by_image(row_start: np.ndarray, world: int, rank: int) -> Tuple[int, int, int, int]:
    """Partition rows at image boundaries so that every rank gets ~n_rows/world rows.
    row_start: [n_images+1] first row of every image.  Returns (img_lo, img_hi, row_lo, row_hi)."""
    n_images = row_start.shape[0] - 1
    n_rows = int(row_start[-1])
    cuts = [0]
    for r in range(1, world):
        target = (r * n_rows) // world
        cuts.append(int(np.searchsorted(row_start, target, side="left")))
    cuts.append(n_images)
    cuts = np.maximum.accumulate(np.minimum(cuts, n_images))
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    return lo, hi, int(row_start[lo]), int(row_start[hi])


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor()."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def merge_keys_hip(device: int, stream_ptr: int, keys, counts, k: int, out_keys, out_count):
    """Global top-k of `world` sorted key lists on the GPU (ssw_topk_merge_dev)."""
    world, stride = keys.shape
    _lib.call("ssw_topk_merge_dev", int(device), ctypes.c_void_p(stream_ptr),
              ctypes.c_void_p(keys.data_ptr()), int(world), int(stride),
              ctypes.c_void_p(counts.data_ptr()), int(k), ctypes.c_void_p(out_keys.data_ptr()),
              ctypes.c_void_p(out_count.data_ptr()))


class ShardedTopK:
    """Exchange + merge step shared by every sharded index.  A shard's local selection leaves its sorted
    composite keys and the pair (count, overflow) in device buffers (ssw_index_result_ptrs); one message per
    rank -- k_max keys [, k_max best-row numbers], then one word holding count | overflow << 32 -- travels in a
    single all-gather.  The overflow flags of all ranks stay on the device (`flags`); `overflowed()` reads them
    (synchronises).

    comm_device: where the collective's tensors live.  None = `device` (backend nccl = RCCL over xGMI: the
    production path).  "cpu" with a gloo group: the message makes a host round trip around the collective."""

    def __init__(self, *, rank: int, world: int, device, image_offset: int, k_max: int,
                 group=None, merge=merge_keys_hip, force_collective: bool = False, with_best: bool = False,
                 comm_device=None):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        self.force_collective = force_collective  # run the all-gather even at world size 1 (RCCL smoke test)
        self.device = device
        self.image_offset = int(image_offset)
        self.group = group
        self.merge = merge
        self.k_max = int(k_max)
        self.with_best = bool(with_best)
        self.msg_len = (2 if with_best else 1) * self.k_max + 1
        dev = device
        self.comm_device = torch.device(comm_device) if comm_device is not None else None
        self.send_buf = torch.zeros(self.msg_len, dtype=torch.int64, device=dev)
        self.send_keys = self.send_buf[:self.k_max]
        self.all_buf = torch.zeros((world, self.msg_len), dtype=torch.int64, device=dev)
        self.all_keys = self.all_buf  # rows are read with stride msg_len
        self.all_counts = torch.zeros(world, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(world, dtype=torch.int64, device=dev)      # overflow flag of every rank, last exchange
        self.flags_seen = torch.zeros(1, dtype=torch.int64, device=dev)    # OR over all exchanges since reset
        self.out_keys = torch.zeros(self.k_max, dtype=torch.int64, device=dev)
        self.out_count = torch.zeros(1, dtype=torch.int32, device=dev)

    # ---- fused form: no elementwise kernels around the collective ---------------------------------------
    def attach(self, dev_index, row_offset: int = 0):
        """have the index's selection write this rank's message itself (ssw_index_set_exchange_target): its last kernel
        then leaves keys - image_offset, best rows + row_offset and count | overflow << 32 in send_buf"""
        assert self.send_buf.is_cuda
        _lib.call("ssw_index_set_exchange_target", dev_index._h, ctypes.c_void_p(self.send_buf.data_ptr()), self.k_max,
                  int(self.with_best), self.image_offset, int(row_offset))
        self._attached = dev_index
        return self

    def use_c_comm(self, unique_id: Optional[bytes] = None):
        """route the all-gather through the library's own entry point (ssw_topk_allgather: ncclAllGather on the current
        stream) instead of torch.distributed.  The 128-byte id comes from rank 0 (ssw_comm_unique_id) and is handed
        to the other ranks through the default process group when there is one."""
        torch = self.torch
        if unique_id is None:
            buf = (ctypes.c_char * 128)()
            if self.rank == 0:
                _lib.call("ssw_comm_unique_id", buf)
            uid = bytes(buf)
            if self.world > 1:
                import torch.distributed as dist
                t = torch.tensor(list(uid), dtype=torch.uint8, device=self.send_buf.device if dist.get_backend(self.group) == "nccl" else "cpu")
                dist.broadcast(t, src=0, group=self.group)
                uid = bytes(t.cpu().tolist())
        else:
            uid = unique_id
        self._comm = ctypes.c_void_p()
        dev_index = self.send_buf.device.index if self.send_buf.is_cuda else 0
        _lib.call("ssw_comm_create", int(dev_index), uid, self.rank, self.world, ctypes.byref(self._comm))
        return self

    def close_c_comm(self):
        if getattr(self, "_comm", None):
            _lib.call("ssw_comm_destroy", self._comm)
            self._comm = None

    def exchange_fused(self, k: int):
        """the attached index has just run topk_dev(q, k): all-gather its message, merge (counts and overflow flags are
        unpacked by the merge kernel) -> (out_keys, out_count).  Two launches + one collective, nothing else."""
        torch = self.torch
        assert getattr(self, "_attached", None) is not None, "attach(dev_index) first"
        if not 1 <= k <= self.k_max:
            raise ValueError(f"k={k} outside [1, k_max={self.k_max}] of this exchange")
        stream_ptr = torch.cuda.current_stream().cuda_stream
        timed = getattr(self, "_coll_events", None) is not None
        if timed:  # HIP events on the stream the collective runs on (bench.py's allgather_us)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if getattr(self, "_comm", None):
            _lib.call("ssw_topk_allgather", self._comm, ctypes.c_void_p(stream_ptr), ctypes.c_void_p(self.send_buf.data_ptr()),
                      ctypes.c_void_p(self.all_buf.data_ptr()), self.msg_len)
        else:
            self.gather()
        if timed:
            e1.record()
            self._coll_events.append((e0, e1))
        _lib.call("ssw_topk_merge_msgs_dev", int(self.all_buf.device.index), ctypes.c_void_p(stream_ptr),
                  ctypes.c_void_p(self.all_buf.data_ptr()), self.world, self.k_max, int(self.with_best), int(k),
                  ctypes.c_void_p(self.out_keys.data_ptr()), ctypes.c_void_p(self.out_count.data_ptr()),
                  ctypes.c_void_p(self.flags.data_ptr()), ctypes.c_void_p(self.flags_seen.data_ptr()))
        return self.out_keys, self.out_count

    def time_collective(self, on: bool = True):
        """bracket every exchange_fused collective with HIP events (two event records per step on the stream)"""
        self._coll_events = [] if on else None

    def collective_us(self):
        """microseconds of each timed collective since time_collective(True) (synchronises)"""
        ev = getattr(self, "_coll_events", None) or []
        self.torch.cuda.synchronize()
        return [1e3 * a.elapsed_time(b) for a, b in ev]

    def pack(self, local_keys, local_count, k: int, image_offset: Optional[int] = None, best_rows=None):
        """local_keys: int64 tensor [>=k] (bit pattern of the u64 keys, local image positions);
        local_count: int32 tensor [1] = count or [2] = (count, overflow); best_rows (with_best): int64 tensor
        [>=k], already global.  Fills this rank's message."""
        torch = self.torch
        assert 1 <= k <= self.k_max
        off = self.image_offset if image_offset is None else int(image_offset)
        # globalise: low 32 bits hold 0xFFFFFFFF - local_id, so subtracting the shard's
        # first image position yields 0xFFFFFFFF - global_id (no borrow: ids < 2^32)
        self.send_keys[:k] = local_keys[:k] - off
        if self.with_best:
            self.send_buf[self.k_max:self.k_max + k] = best_rows[:k]
        word = local_count[:1].to(torch.int64)
        if local_count.shape[0] > 1:
            word = word | (local_count[1:2].to(torch.int64) << 32)
        self.send_buf[self.msg_len - 1:] = word
        return self.send_buf

    def pack_empty(self):
        """a rank without images still takes part in the collective: count 0"""
        self.send_buf[self.msg_len - 1:] = 0
        return self.send_buf

    def gather(self):
        if self.world > 1 or self.force_collective:
            import torch.distributed as dist
            # flat (concatenating) form: accepted by both RCCL and gloo
            if self.comm_device is None or self.comm_device == self.send_buf.device:
                dist.all_gather_into_tensor(self.all_buf.view(-1), self.send_buf, group=self.group)
            else:
                out = self.torch.empty(self.world * self.msg_len, dtype=self.torch.int64, device=self.comm_device)
                dist.all_gather_into_tensor(out, self.send_buf.to(self.comm_device), group=self.group)
                self.all_buf.view(-1).copy_(out)
        else:
            self.all_buf[0] = self.send_buf

    def merge_gathered(self, k: int):
        """global top-k of the gathered messages (all_buf) on every rank -> (out_keys[k_max], out_count[1])"""
        torch = self.torch
        words = self.all_buf[:, self.msg_len - 1]
        self.all_counts.copy_(words & 0xFFFFFFFF)
        self.flags.copy_(words >> 32)
        self.flags_seen |= self.flags.max()
        stream_ptr = torch.cuda.current_stream().cuda_stream if self.all_keys.is_cuda else 0
        dev_index = self.all_keys.device.index if self.all_keys.is_cuda else -1
        self.merge(dev_index, stream_ptr, self.all_keys, self.all_counts, k, self.out_keys,
                   self.out_count)
        return self.out_keys, self.out_count

    def exchange(self, local_keys, local_count, k: int, best_rows=None):
        self.pack(local_keys, local_count, k, best_rows=best_rows)
        self.gather()
        return self.merge_gathered(k)

    def best_rows_of(self, merged_keys: np.ndarray) -> np.ndarray:
        """the best-row number every rank sent along with each of the merged keys (keys are unique)"""
        assert self.with_best
        buf = self.all_buf.cpu().numpy()
        counts = (buf[:, self.msg_len - 1] & 0xFFFFFFFF).astype(np.int64)
        keys = np.concatenate([buf[r, :counts[r]] for r in range(self.world)]).view(np.uint64)
        rows = np.concatenate([buf[r, self.k_max:self.k_max + counts[r]] for r in range(self.world)])
        order = np.argsort(keys)
        at = np.searchsorted(keys[order], np.asarray(merged_keys, dtype=np.uint64))
        assert np.array_equal(keys[order][at], merged_keys)
        return rows[order][at]

    def overflowed(self):
        """ranks whose last local selection overflowed its fast path (host read: synchronises).  Every rank
        sees the same list, so all of them can agree to repeat the exchange after the deep selection."""
        return [int(r) for r in self.torch.nonzero(self.flags.cpu()).reshape(-1)]

    def assert_no_overflow_seen(self):
        """for callers of the asynchronous form: fail loudly if any exchange since the last reset carried an
        overflow flag (the merged keys of that query were not the exact top-k)"""
        if int(self.flags_seen.cpu().item()) != 0:
            raise RuntimeError("sharded top-k: a shard's fast selection overflowed (mass ties / duplicated vectors); "
                               "use the synchronous topk(), which reruns the exact deep selection")

    def reset_overflow_seen(self):
        self.flags_seen.zero_()


class ShardedSyntheticIndex:
    """BASELINE config C4: N_total x dim synthetic rows, one vector per image, row-sharded
    over the ranks of the default process group; each rank generates its slice in place."""

    def __init__(self, n_total: int, dim: int, seed: int, rank: int, world: int,
                 local_device: int, k_max: int = 128, group=None, force_collective: bool = False, comm_device=None):
        import torch
        from .device_index import DeviceIndex
        self.torch = torch
        self.n_total, self.dim = int(n_total), int(dim)
        self.rank, self.world = rank, world
        self.row_lo, self.row_hi = shard_bounds(self.n_total, world, rank)
        self.n_local = self.row_hi - self.row_lo
        self.device = torch.device("cuda", local_device)
        self.local = DeviceIndex.synthetic(self.n_local, dim, seed=seed, first_row=self.row_lo,
                                           device=local_device)
        # run the library's kernels on torch's current stream so the collective and the
        # merge are ordered behind the scan without host synchronisation
        self.local.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        keys_ptr, count_ptr, _ = self.local.result_ptrs()
        self.local_keys = torch.as_tensor(_DevArray(keys_ptr, (_lib.SSW_MAX_TOPK,), "<i8"),
                                          device=self.device)
        self.local_count = torch.as_tensor(_DevArray(count_ptr, (2,), "<i4"), device=self.device)  # count, overflow
        self.xchg = ShardedTopK(rank=rank, world=world, device=self.device,
                                image_offset=self.row_lo, k_max=k_max, group=group,
                                force_collective=force_collective, comm_device=comm_device)
        # the selection writes the message itself, the merge unpacks it: scan -> select -> all-gather -> merge
        self.xchg.attach(self.local)
        if os.environ.get("SSW_C_COMM"):  # the collective through the library's own RCCL entry point
            self.xchg.use_c_comm()

    def topk_async(self, q_dev_ptr: int, k: int):
        """scan + local select + all-gather + merge, all enqueued on the current stream.  The overflow flags
        travel with the message; check `xchg.assert_no_overflow_seen()` after synchronising."""
        if not 1 <= k <= self.xchg.k_max:  # ahead of the selection, which writes k words of the message
            raise ValueError(f"k={k} outside [1, k_max={self.xchg.k_max}] of this index's exchange")
        self.local.topk_dev(q_dev_ptr, k)
        return self.xchg.exchange_fused(k)

    def topk(self, q_dev_ptr: int, k: int):
        from .device_index import decode_keys
        keys, count = self.topk_async(q_dev_ptr, k)
        self.torch.cuda.synchronize(self.device)
        over = self.xchg.overflowed()
        if over:  # the same list on every rank: the overflowing ones redo their selection exactly, all re-exchange
            if self.rank in over:
                self.local.select_deep_dev(k)
            if self.rank not in over:  # (a repaired shard's deep selection rewrote its message; the others re-send theirs)
                pass
            keys, count = self.xchg.exchange_fused(k)
            self.torch.cuda.synchronize(self.device)
            assert not self.xchg.overflowed()
            self.xchg.reset_overflow_seen()
        c = int(count.item())
        imgs, scores = decode_keys(keys[:c].cpu().numpy().view(np.uint64))
        return imgs, scores

    def close(self):
        self.xchg.close_c_comm()
        self.local.close()
