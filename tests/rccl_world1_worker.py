"""Worker of tests/test_sharded_gpu.py::test_exchange_over_rccl_world_size_1 (launched by torch.distributed.run).
Runs the sharded index path of bench.py with a real `nccl` (RCCL) process group of world size 1, forcing the
collective branch that world == 1 normally skips, and checks the result against the plain device index."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from oracle import seesaw_oracle as orc
    from seesaw_amd.device_index import DeviceIndex
    from seesaw_amd.sharded import ShardedSyntheticIndex
    n, k = 200_000, 100
    real_world = dist.get_world_size()
    # force_collective: take the all_gather_into_tensor branch that world == 1 normally skips
    index = ShardedSyntheticIndex(n, 512, 5, dist.get_rank(), real_world, local_rank, k_max=128, force_collective=True)
    q = orc.synth_query(3)
    q_dev = torch.from_numpy(q).cuda()
    imgs, scores = index.topk(q_dev.data_ptr(), k)
    plain = DeviceIndex.synthetic(n, 512, seed=5)
    p_imgs, p_scores, _ = plain.topk(q, k)
    assert np.array_equal(imgs, p_imgs), "RCCL path differs from the plain index"
    assert np.array_equal(scores.view(np.uint32), p_scores.view(np.uint32))
    # the same query with the collective issued by the library itself (ssw_comm_create / ssw_topk_allgather: RCCL bound
    # by dlopen, ncclAllGather on the current stream) instead of torch.distributed
    index.xchg.use_c_comm()
    imgs2, scores2 = index.topk(q_dev.data_ptr(), k)
    assert np.array_equal(imgs2, p_imgs) and np.array_equal(scores2.view(np.uint32), p_scores.view(np.uint32)), \
        "ssw_topk_allgather path differs from the plain index"
    index.xchg.close_c_comm()
    # a bare collective on device tensors too, so a failure points at RCCL rather than at the index
    t = torch.arange(8, dtype=torch.int64, device="cuda")
    out = torch.empty(8 * real_world, dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(out, t)
    assert torch.equal(out[:8], t)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK", flush=True)


if __name__ == "__main__":
    main()
