"""Parity at BASELINE.json's full sizes (configs C2 = 1 M rows, C4 = 100 M rows on one GPU).

C2 is small enough for the CPU oracle to finish in seconds, so it is compared bit for bit.
C4 (204.8 GB of rows) is checked through properties that do not need a full CPU scan:
  * every returned score is bit-identical to the oracle's score of that row (rows regenerated
    one by one from the counter RNG);
  * the result is sorted by the reference's order (score descending, image id ascending), ids distinct;
  * no row of a 20 000-row random sample beats the k-th score without being in the result;
  * top-2k == top-k followed by top-k with the first k excluded (the stateful-exclusion path of
    query_interface.py:34-49);
  * the whole-index result equals the merge of the 8 image-range shards' local results
    (the C4 sharding of SURVEY section 8e, here run shard after shard on one GPU).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def DeviceIndex():
    from seesaw_amd.device_index import DeviceIndex
    return DeviceIndex


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_c2_one_million_rows_bit_exact(DeviceIndex, oracle):
    n, k, seed = 1_000_000, 100, 2024
    idx = DeviceIndex.synthetic(n, 512, seed=seed)
    X = oracle.synth_rows(seed, 0, n, 512)
    rng = np.random.default_rng(5)
    excluded = rng.choice(n, size=1000, replace=False).tolist()
    for qi, ex in ((1, []), (2, excluded)):
        q = oracle.synth_query(qi)
        ref_scores = oracle.scores_kernel_order(X, q)
        imgs, scores, rows = idx.topk(q, k, excluded=ex)
        o_imgs, o_scores, o_rows = oracle.topk_images_tiebreak(ref_scores, None, n, ex, k)
        assert np.array_equal(imgs, o_imgs)
        assert np.array_equal(bits(scores), bits(o_scores))
        assert np.array_equal(rows, o_rows)
        # the whole score vector, not just the winners
        assert np.array_equal(bits(idx.scores(q)), bits(ref_scores))
    idx.close()


def _oracle_scores_of_rows(oracle, seed, rows, q):
    out = np.empty(len(rows), dtype=np.float32)
    for i, r in enumerate(rows):
        out[i] = oracle.scores_kernel_order(oracle.synth_rows(seed, int(r), 1, 512), q)[0]
    return out


def _merge(parts, k):
    """Reference order over (score desc, image id asc) of the concatenated local results."""
    imgs = np.concatenate([p[0] for p in parts])
    scores = np.concatenate([p[1] for p in parts])
    order = np.lexsort((imgs, -scores.astype(np.float64)))[:k]
    return imgs[order], scores[order]


def test_c4_hundred_million_rows_properties(DeviceIndex, oracle):
    import torch
    n, k, seed = 100_000_000, 100, 2024
    free, _total = torch.cuda.mem_get_info(0)  # (seesaw_amd._lib loads torch's HIP runtime first)
    if free < n * 2048 + (8 << 30):
        pytest.skip("needs the 288 GB of an MI355X")
    q = oracle.synth_query(11)
    idx = DeviceIndex.synthetic(n, 512, seed=seed)
    imgs, scores, rows = idx.topk(q, k)
    assert imgs.shape == (k,) and np.array_equal(imgs, rows)  # one row per image in C4
    # (1) returned scores are the oracle's scores of those rows
    assert np.array_equal(bits(scores), bits(_oracle_scores_of_rows(oracle, seed, rows, q)))
    # (2) reference order, distinct ids
    assert len(set(imgs.tolist())) == k
    key = list(zip((-scores.astype(np.float64)).tolist(), imgs.tolist()))
    assert key == sorted(key)
    # (3) sampled completeness
    sample = np.random.default_rng(17).integers(0, n, size=20000)
    s_scores = _oracle_scores_of_rows(oracle, seed, sample, q)
    inside = set(imgs.tolist())
    for r, s in zip(sample.tolist(), s_scores.tolist()):
        if s > scores[-1] or (s == scores[-1] and r < imgs[-1]):
            assert r in inside
    # (3b) COMPLETE coverage (VERDICT r3 weak #1e: the sample above sees 20 000 of 10^8 rows).  Every row's score against an
    # independent implementation -- torch.mv (rocBLAS) over the resident matrix, in chunks of 4 M rows -- inside the f32
    # rounding band of a 512-term dot product of unit vectors; and the selection against the whole resident score vector:
    # exactly k - 1 rows rank before the k-th returned one under (score desc, row asc), all of them returned.
    from seesaw_amd.sharded import _DevArray
    vec_ptr, score_ptr = idx.device_ptrs()
    dev = torch.device("cuda", 0)
    Xd = torch.as_tensor(_DevArray(vec_ptr, (n, 512), "<f4"), device=dev)
    sd = torch.as_tensor(_DevArray(score_ptr, (n,), "<f4"), device=dev)
    qd = torch.from_numpy(q).to(dev)
    worst = 0.0
    for lo in range(0, n, 4_000_000):
        hi = min(n, lo + 4_000_000)
        worst = max(worst, float((torch.mv(Xd[lo:hi], qd) - sd[lo:hi]).abs().max()))
    assert worst <= 512 * 2.0 ** -24 * 2, worst               # |x| = |q| = 1: sum |x_i q_i| <= 1, 512 roundings of 2^-24
    kth_score, kth_row = float(scores[-1]), int(imgs[-1])
    before = int((sd > kth_score).sum()) + int(((sd == kth_score) & (torch.arange(n, device=dev) < kth_row)).sum())
    assert before == k - 1, before
    top = torch.topk(sd, k).values.cpu().numpy()
    assert np.array_equal(bits(np.sort(top)[::-1]), bits(scores))   # the k largest values of the vector, as a multiset
    del Xd, sd
    # (4) top-2k == top-k ++ top-k after excluding the first k
    imgs2k, scores2k, _ = idx.topk(None, 2 * k)
    nxt_imgs, nxt_scores, _ = idx.topk(None, k, excluded=imgs.tolist())
    assert np.array_equal(imgs2k[:k], imgs) and np.array_equal(imgs2k[k:], nxt_imgs)
    assert np.array_equal(bits(scores2k[k:]), bits(nxt_scores))
    imgs1k, scores1k, _ = idx.topk(None, 1024)
    idx.close()
    del idx
    # (5) 8 image-range shards of 12.5 M rows, scanned one after the other, merge to the same answer -- on the host, and
    # (6) (VERDICT r3 #5b) through the N > 1 step's own pipeline: every shard's selection writes its exchange message
    # itself (k_final with an exchange target: keys made global, best rows, count | overflow), the eight REAL messages are
    # stacked as the all-gather would leave them and ssw_topk_merge_msgs_dev merges them, at k = 100 and k = 1024
    import ctypes
    from seesaw_amd import _lib
    from seesaw_amd.device_index import decode_keys
    from seesaw_amd.sharded import ShardedTopK
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    q_dev = torch.from_numpy(q).cuda()
    parts = []
    per = n // 8
    k_max = 1024
    gathered = {kk: ShardedTopK(rank=0, world=8, device=dev, image_offset=0, k_max=k_max, with_best=True) for kk in (k, 1024)}
    for r in range(8):
        shard = DeviceIndex.synthetic(per, 512, seed=seed, first_row=r * per)
        li, ls, _ = shard.topk(q, k)
        parts.append((li + r * per, ls))
        shard.set_stream(stream)
        one = ShardedTopK(rank=r, world=8, device=dev, image_offset=r * per, k_max=k_max, with_best=True)
        one.attach(shard, row_offset=r * per)
        for kk in (k, 1024):
            shard.topk_dev(q_dev.data_ptr() if kk == k else 0, kk)  # (the second selection reuses the resident scores)
            torch.cuda.synchronize()
            assert int(one.send_buf[-1].item()) == kk                # count kk, overflow flag clear
            gathered[kk].all_buf[r] = one.send_buf
        shard.close()
    m_imgs, m_scores = _merge(parts, k)
    assert np.array_equal(m_imgs, imgs) and np.array_equal(bits(m_scores), bits(scores))
    for kk, want_imgs, want_scores in ((k, imgs, scores), (1024, imgs1k, scores1k)):
        x = gathered[kk]
        _lib.call("ssw_topk_merge_msgs_dev", 0, ctypes.c_void_p(stream), ctypes.c_void_p(x.all_buf.data_ptr()), 8, k_max, 1, kk,
                  ctypes.c_void_p(x.out_keys.data_ptr()), ctypes.c_void_p(x.out_count.data_ptr()),
                  ctypes.c_void_p(x.flags.data_ptr()), ctypes.c_void_p(x.flags_seen.data_ptr()))
        torch.cuda.synchronize()
        assert int(x.out_count.item()) == kk and int(x.flags_seen.item()) == 0
        got_imgs, got_scores = decode_keys(x.out_keys[:kk].cpu().numpy().view(np.uint64))
        assert np.array_equal(got_imgs, want_imgs) and np.array_equal(bits(got_scores), bits(want_scores)), kk
        assert np.array_equal(x.best_rows_of(x.out_keys[:kk].cpu().numpy().view(np.uint64)), want_imgs)  # one row per image


@pytest.mark.parametrize("shape", ["gaussian", "sorted", "clustered", "hidden_from_the_sample", "mostly_excluded"])
def test_sampled_threshold_selection_is_exact(DeviceIndex, lab_build, shape):
    """from 2^24 values on the selection takes its threshold from a 1-in-16 sample (blocks of 16 neighbours); whatever
    the sample sees, the answer is the exact top-k in (score desc, position asc) order -- a sample that misleads
    (here: every large value sits where no sampled block looks) ends in the deep path instead"""
    import torch
    from seesaw_amd import _lib
    n = (1 << 24) + 12345
    free, _total = torch.cuda.mem_get_info(0)
    if free < n * 2048 + (8 << 30):
        pytest.skip("needs 40 GB of device memory")
    rng = np.random.default_rng(len(shape))
    s = (rng.standard_normal(n) * 0.044).astype(np.float32)
    excluded = rng.choice(n, 500, replace=False)
    if shape == "sorted":
        s.sort()
    elif shape == "clustered":  # neighbours share most of their score, like the tiles of one image
        s = (np.repeat(rng.standard_normal(n // 16 + 1), 16)[:n] * 0.04 + s * 0.1).astype(np.float32)
    elif shape == "hidden_from_the_sample":
        pos = np.arange(n)
        s[(pos % 256) >= 16] += np.float32(1.0)  # sampled blocks are elements 0..15 of every 256
    elif shape == "mostly_excluded":
        excluded = np.setdiff1d(np.arange(n), rng.choice(n, 3000, replace=False))
    idx = DeviceIndex.synthetic(n, 512, seed=1)
    try:
        idx.load_scores(s)
        sx = s.astype(np.float64)
        sx[excluded] = -np.inf
        for k in (1, 100, 2048):
            part = np.argpartition(-sx, k + 64)[: k + 64]
            order = part[np.lexsort((part, -sx[part]))][:k]
            if shape == "mostly_excluded":
                order = order[np.isfinite(sx[order])]
            for flags in (3, 1):
                _lib.call("ssw_tune_topk", flags)
                imgs, scores, rows = idx.topk(None, k, excluded=excluded)
                assert np.array_equal(imgs, order), (shape, k, flags)
                assert np.array_equal(bits(scores), bits(s[order]))
    finally:
        _lib.call("ssw_tune_topk", 3)
        idx.close()


def test_sampled_threshold_at_the_bottom_of_a_level1_bin_takes_level2(DeviceIndex):
    """in sampled mode (m >= 2^24) the level-2 decision was taken on the SAMPLE's counts, which stand for 64
    times as many values.  Here ~19 000 values fill one 12-bit bin above everything else: the sample holds ~300 of them
    (below the 1024 that used to trigger level 2), so the 12-bit prefix alone collected all 19 000 -- beyond the 8192
    the final selection takes: overflow word raised, deep rerun on every query.  With the decision on the scaled count
    the second level runs, a few thousand candidates come out and nothing overflows."""
    import torch
    from seesaw_amd.sharded import _DevArray
    n = (1 << 24) + 4097
    free, _total = torch.cuda.mem_get_info(0)
    if free < n * 2048 + (8 << 30):
        pytest.skip("needs 40 GB of device memory")
    rng = np.random.default_rng(5)
    s = (rng.random(n) * 0.12).astype(np.float32)                       # bins below
    top = rng.choice(n, 19000, replace=False)
    s[top] = (0.125 + rng.random(19000) * 0.015).astype(np.float32)     # inside the bin [0.125, 0.140625)
    idx = DeviceIndex.synthetic(n, 512, seed=1)
    try:
        idx.load_scores(s)
        _, count_ptr, _ = idx.result_ptrs()
        count = torch.as_tensor(_DevArray(count_ptr, (2,), "<i4"), device=torch.device("cuda", 0))
        for k in (50, 100, 1024):
            idx.topk_dev(0, k)
            idx.sync()
            c, ovf = (int(v) for v in count.cpu())
            assert (c, ovf) == (k, 0), (k, c, ovf)
            imgs, scores, _ = idx.topk_fetch(k)
            part = np.argpartition(-s, k + 64)[: k + 64]
            order = part[np.lexsort((part, -s[part]))][:k]
            assert np.array_equal(imgs, order)
            assert np.array_equal(bits(scores), bits(s[order]))
    finally:
        idx.close()
