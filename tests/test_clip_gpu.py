"""GPU: CLIP ViT-B/32 towers (bf16 MFMA) against the in-container transformers.CLIPModel
(f32, torch-CPU) with seeded random-init weights.

The reference holds no fixture for CLIP and delegates its arithmetic to `transformers`
(parity unpinned by the reference); the pin is the HF implementation itself.
Tolerance: the product path multiplies in bf16 (f32 accumulate), so the bar is the one
BASELINE.md section 3 states for the MFMA path -- cosine >= 0.999 against the f32 oracle on the
L2-normalised 512-d output -- plus an absolute bound on the components."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COS_MIN = 0.999
ABS_MAX = 5e-3  # per component of a unit vector (typical component ~0.044)


@pytest.fixture(scope="module")
def models():
    import torch
    import transformers
    from seesaw_amd.models.clip import ClipModel
    torch.manual_seed(1234)
    hf = transformers.CLIPModel(transformers.CLIPConfig()).eval()
    return hf, ClipModel.from_hf(hf)


def _unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def test_image_tower_vs_hf(models):
    import torch
    hf, ours = models
    torch.manual_seed(0)
    x = torch.randn(13, 3, 224, 224)  # 13 tiles = one COCO-shape image
    with torch.inference_mode():
        ref = hf.get_image_features(pixel_values=x)
        ref = ref.pooler_output if hasattr(ref, "pooler_output") else ref
        ref = torch.nn.functional.normalize(ref, dim=1).numpy()
    got = ours.embed_image(x.numpy(), normalize=True)
    assert got.shape == (13, 512)
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    cos = (got * ref).sum(1)
    assert cos.min() >= COS_MIN, cos
    assert np.abs(got - ref).max() <= ABS_MAX, np.abs(got - ref).max()
    # different inputs must give different embeddings (no collapsed path)
    assert np.abs(got[0] - got[1]).max() > 1e-3


def test_residual_stream_precisions_vs_hf(models):
    """The tile path holds both towers' residual rows in f32 by default, in bf16 on request
    (ssw_clip_set_option, per handle): both arithmetic forms of both towers meet the bar, and the switch is live."""
    import torch
    from seesaw_amd import _lib
    hf, ours = models
    torch.manual_seed(5)
    x = torch.randn(6, 3, 224, 224)
    rng = np.random.default_rng(11)
    ids = rng.integers(0, 49405, size=(4, 40)).astype(np.int64)
    ids[:, 0] = 49406
    ids[:, 39] = 49407
    with torch.inference_mode():
        ref_i = hf.get_image_features(pixel_values=x)
        ref_i = _unit((ref_i.pooler_output if hasattr(ref_i, "pooler_output") else ref_i).numpy())
        ref_t = hf.get_text_features(input_ids=torch.from_numpy(ids))
        ref_t = _unit((ref_t.pooler_output if hasattr(ref_t, "pooler_output") else ref_t).numpy())
    got = {}
    try:
        for flags in (0, 3):  # 0: f32 rows in both towers (default); 3: bf16 rows in both
            ours.set_rows(image_bf16=bool(flags & 1), text_bf16=bool(flags & 2))
            got[flags] = (ours.embed_image(x.numpy(), normalize=True), _unit(ours.embed_text(ids.astype(np.int32), normalize=False)))
    finally:
        ours.set_rows()
    for flags, (gi, gt) in got.items():
        assert (gi * ref_i).sum(1).min() >= COS_MIN and (gt * ref_t).sum(1).min() >= COS_MIN, flags
        assert np.abs(gi - ref_i).max() <= ABS_MAX and np.abs(gt - ref_t).max() <= ABS_MAX, flags
    for a, b in zip(got[0], got[3]):
        assert 0 < np.abs(a - b).max() <= 3e-3
    with pytest.raises(_lib.SeesawHipError):
        ours.set_option(5, True)  # beyond SSW_CLIP_OPT_FULL_LAST_LAYER


def test_image_attention_forms_give_the_same_bits(models):
    """attention_rows64 (K / Q / V rows fetched coalesced, fragments out of LDS, the output tile stored row-major) against
    attention_mfma<4, 2> (fragments straight from memory): same fragments, same MFMA order -- identical embeddings."""
    _, ours = models
    x = np.random.default_rng(21).standard_normal((9, 3, 224, 224), dtype=np.float32)
    try:
        ours.set_option(ours.OPT_ATTN_OUT_UNFUSED, True)  # both forms are attention launches of their own
        a = ours.embed_image(x, normalize=False)
        ours.set_option(ours.OPT_ATTN_DIRECT, True)
        b = ours.embed_image(x, normalize=False)
    finally:
        ours.set_option(ours.OPT_ATTN_DIRECT, False)
        ours.set_option(ours.OPT_ATTN_OUT_UNFUSED, False)
    assert a.tobytes() == b.tobytes()


def test_fused_attention_outprojection_against_the_two_launches(models):
    """round 4: attention + out-projection + residual + statistics in one launch per layer (attn_out.hip).  The attention
    tiles and the product's MFMA order are those of the two launches; only the grouping of the LayerNorm partial sums
    differs (two partial pairs a row instead of six), so the embeddings agree to f32 rounding of those sums -- in both
    residual-row precisions."""
    _, ours = models
    x = np.random.default_rng(22).standard_normal((7, 3, 224, 224), dtype=np.float32)
    try:
        for f32_rows in (False, True):
            ours.set_rows(image_bf16=not f32_rows)
            ours.set_option(ours.OPT_ATTN_OUT_UNFUSED, False)
            a = ours.embed_image(x, normalize=True)
            ours.set_option(ours.OPT_ATTN_OUT_UNFUSED, True)
            b = ours.embed_image(x, normalize=True)
            # (measured 9e-4 with bf16 rows: a 1e-7 change of a row's mean flips bf16 roundings of later rows, one ulp =
            #  7.8e-3 at |x| ~ 1.5, and the flips travel; the two text paths sit as far apart, test_single_query_...)
            assert np.abs(a - b).max() <= 1.5e-3, (f32_rows, np.abs(a - b).max())
            assert (a * b).sum(1).min() >= 0.99995
    finally:
        ours.set_rows()
        ours.set_option(ours.OPT_ATTN_OUT_UNFUSED, False)


@pytest.mark.parametrize("L", [8, 77])
def test_text_tower_vs_hf(models, L):
    import torch
    hf, ours = models
    rng = np.random.default_rng(L)
    B = 5
    ids = rng.integers(0, 49405, size=(B, L)).astype(np.int64)
    ids[:, 0] = 49406
    eos_at = rng.integers(2, L, size=B)
    eos_at[0] = L - 1
    for b in range(B):
        ids[b, eos_at[b]] = 49407  # features are read at the first EOS; tokens after it are padding-like
    with torch.inference_mode():
        ref = hf.get_text_features(input_ids=torch.from_numpy(ids))
        ref = (ref.pooler_output if hasattr(ref, "pooler_output") else ref).numpy()
    got = ours.embed_text(ids.astype(np.int32), normalize=False)
    cos = (_unit(got) * _unit(ref)).sum(1)
    assert cos.min() >= COS_MIN, cos
    assert np.abs(_unit(got) - _unit(ref)).max() <= ABS_MAX
    assert np.abs(np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1) - 1).max() < 2e-2


@pytest.mark.parametrize("B,L", [(1, 3), (1, 8), (1, 17), (1, 32), (2, 16), (3, 5), (1, 33)])
def test_single_query_text_path_vs_hf(models, B, L):
    """up to 32 rows (one short query) go through the skinny kernels (skinny_linear: layer norm fused, K split over the
    waves, five launches a layer); (1, 33) is the first shape back on the tile kernels.  Same bar as the batched
    path, and the two paths agree on the same query far inside that bar."""
    import torch
    hf, ours = models
    rng = np.random.default_rng(100 * B + L)
    ids = rng.integers(0, 49405, size=(B, L)).astype(np.int64)
    ids[:, 0] = 49406
    ids[:, L - 1] = 49407
    if L > 4:
        ids[B - 1, L - 2] = 49407  # an earlier EOS in the last row: pooled there
    with torch.inference_mode():
        ref = hf.get_text_features(input_ids=torch.from_numpy(ids))
        ref = (ref.pooler_output if hasattr(ref, "pooler_output") else ref).numpy()
    got = ours.embed_text(ids.astype(np.int32), normalize=False)
    cos = (_unit(got) * _unit(ref)).sum(1)
    assert cos.min() >= COS_MIN, cos
    assert np.abs(_unit(got) - _unit(ref)).max() <= ABS_MAX
    assert np.abs(np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1) - 1).max() < 2e-2
    # the same rows inside a batch of 40+ rows take the tile kernels
    reps = -(-41 // (B * L))
    tiled = ours.embed_text(np.tile(ids, (reps + 1, 1)).astype(np.int32), normalize=False)[:B]
    assert (_unit(got) * _unit(tiled)).sum(1).min() >= 0.99995
    assert np.abs(_unit(got) - _unit(tiled)).max() <= 1e-3


def test_a_texts_vector_does_not_depend_on_the_texts_that_share_its_call(models):
    """the text tower's fc2 is split over K; the split count used to come from the row count and the CU count,
    so a query's vector depended (1e-7) on its batch.  The split is now a function of the product's shape alone: the same
    sequences embedded 2, 8, 16 and 24 to a call (all on the tile kernels) give the same bytes.  What remains, stated and
    bounded: the final LayerNorm + projection of at most 32 pooled rows is the skinny kernel, of more rows the tile GEMM
    (f32 sums in another order): 48 sequences a call differ from the same sequences 24 a call by <= 5e-6 on the
    unnormalised features (measured 1.4e-6)."""
    _, ours = models
    rng = np.random.default_rng(11)
    ids = rng.integers(0, 49405, size=(48, 77)).astype(np.int32)
    ids[:, 0] = 49406
    ids[:, -1] = 49407
    per_call = {n: np.concatenate([ours.embed_text(ids[a:a + n], normalize=False) for a in range(0, 48, n)]) for n in (2, 8, 16, 24, 48)}
    assert per_call[2].tobytes() == per_call[8].tobytes() == per_call[16].tobytes() == per_call[24].tobytes()
    assert np.abs(per_call[48] - per_call[24]).max() <= 5e-6


def test_text_pooling_legacy_eos_token_id_2():
    """The published openai/clip-vit-* configs still say text_config.eos_token_id = 2; transformers then pools
    at argmax(input_ids) (modeling_clip.py, CLIPTextTransformer.forward).  A tokenised string never contains
    id 2, so pooling at 'the first 2' would read the BOS row for every query."""
    import torch
    import transformers
    from seesaw_amd.models.clip import ClipModel
    cfg = transformers.CLIPConfig()
    cfg.text_config.eos_token_id = 2
    torch.manual_seed(77)
    hf = transformers.CLIPModel(cfg).eval()
    assert hf.text_model.eos_token_id == 2
    ours = ClipModel.from_hf(hf)
    assert ours.eos_token_id == 2
    rng = np.random.default_rng(5)
    B, L = 6, 20
    ids = rng.integers(3, 49405, size=(B, L)).astype(np.int64)
    ids[:, 0] = 49406
    eot_at = rng.integers(2, L, size=B)
    for b in range(B):
        ids[b, eot_at[b]] = 49407
        ids[b, eot_at[b] + 1:] = 0  # padding after the end-of-text token
    with torch.inference_mode():
        ref = hf.get_text_features(input_ids=torch.from_numpy(ids))
        ref = (ref.pooler_output if hasattr(ref, "pooler_output") else ref).numpy()
    got = ours.embed_text(ids.astype(np.int32), normalize=False)
    cos = (_unit(got) * _unit(ref)).sum(1)
    assert cos.min() >= COS_MIN, cos
    assert np.abs(_unit(got) - _unit(ref)).max() <= ABS_MAX
    # different strings -> different vectors (the BOS row would give identical ones)
    assert np.abs(_unit(got)[0] - _unit(got)[1]).max() > 1e-3
    ours.close()


def test_text_without_eos_is_an_error(models):
    from seesaw_amd import _lib
    _, ours = models
    ids = np.full((2, 8), 11, dtype=np.int32)
    ids[0, 5] = 49407
    with pytest.raises(_lib.SeesawHipError, match="end-of-text"):
        ours.embed_text(ids)


def test_embedding_wrappers_and_batch_chunking(models):
    from seesaw_amd.models.embeddings import HGWrapper, ImageEmbedding
    hf, ours = models
    emb = HGWrapper(model=ours)
    v = emb.from_string(string="a dog")
    assert v.shape == (1, 512) and emb.from_string(string="a dog") is v  # cached per string
    assert np.abs(emb.from_string(string="a cat") - v).max() > 1e-4
    rng = np.random.default_rng(0)
    x = rng.standard_normal((300, 3, 224, 224)).astype(np.float32)  # > one 256-image chunk
    out = ImageEmbedding(model=ours)(preprocessed_image=x)
    assert out.shape == (300, 512)
    again = ours.embed_image(x[280:290])
    assert np.array_equal(out[280:290], again)  # batch position does not change the result


def test_u8_tiles_equal_host_batch_tx_path(models):
    """ssw_clip_embed_tiles_u8 fuses batch_tx (multiscale_tools.py:167-183) into the patch gather: the
    bf16 patches, hence the embeddings, are bit-identical to normalising on the host first."""
    import pandas as pd
    from seesaw_amd.indices.multiscale.multiscale_tools import batch_tx
    _, ours = models
    tiles = np.random.default_rng(4).integers(0, 256, size=(21, 224, 224, 3), dtype=np.uint8)
    host = np.stack(batch_tx(pd.DataFrame({"tile": list(tiles)})).tile.values)
    a = ours.embed_image(host, normalize=True)
    b = ours.embed_tiles_u8(tiles, normalize=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_create_multiscale_index_round_trip(models, tmp_path):
    """image bytes -> pyramid tiles -> GPU embedding -> the reference's on-disk layout -> from_path ->
    a text query over it (multiscale_tools.py:225-261, multiscale_index.py:235-269)."""
    import io
    import json
    import pandas as pd
    import PIL.Image
    from seesaw_amd.indices.multiscale.multiscale_index import MultiscaleIndex
    from seesaw_amd.indices.multiscale.multiscale_tools import (create_multiscale_index, generate_multiscale_tiling,
                                                                read_vector_parquet)
    from seesaw_amd.models.embeddings import load_clip
    rng = np.random.default_rng(9)
    rows = []
    for dbidx, (w, h) in enumerate([(640, 480), (224, 224), (500, 375), (448, 448)]):
        buf = io.BytesIO()
        PIL.Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(buf, format="PNG")
        rows.append({"dbidx": dbidx, "file_path": f"img{dbidx}.png", "bytes": buf.getvalue()})
    image_rows = pd.DataFrame(rows)
    model = load_clip("synthetic:random-init")  # explicit opt-in; from_path reloads the same
    path = create_multiscale_index(image_rows=image_rows, dataset_path=str(tmp_path), index_name="multiscale",
                                   model=model, model_path="synthetic:random-init")
    info = json.load(open(f"{path}/info.json"))
    assert info["constructor"] == "seesaw.indices.multiscale.multiscale_index.MultiscaleIndex"
    meta, vecs = read_vector_parquet(f"{path}/vectors.sorted.cached")
    assert list(meta.columns[:3]) == ["dbidx", "file_path", "patch_id"]
    for c in ("zoom_level", "x1", "y1", "x2", "y2", "scale_factor", "max_zoom_level"):
        assert c in meta.columns
    assert vecs.dtype == np.float32 and vecs.shape == (meta.shape[0], 512)
    assert np.abs(np.linalg.norm(vecs, axis=1) - 1).max() < 1e-5
    assert meta.dbidx.is_monotonic_increasing
    counts = meta.groupby("dbidx").size().tolist()
    assert counts[0] == 13 and counts[1] == 1  # 640x480 -> 13 tiles, 224x224 -> 1
    # the stored vector of a tile is the embedding of that tile
    im0 = PIL.Image.open(io.BytesIO(rows[0]["bytes"])).convert("RGB")
    t0 = generate_multiscale_tiling(im0, factor=0.5, tile_size=224, min_tile_size=224)
    again = model.embed_tiles_u8(np.stack(t0.tile.values), normalize=True)
    assert np.array_equal(again.view(np.uint32), vecs[:13].view(np.uint32))
    idx = MultiscaleIndex.from_path(path, use_vec_index=False)
    assert idx.vectors.shape == vecs.shape and len(idx) == 4
    res = idx.query(vector=idx.string2vec("a photo"), topk=2, shortlist_size=4, agg_method="plain_score",
                    aug_larger="greater", rescore_method="plain")
    assert len(res["dbidxs"]) == 2 and set(res["dbidxs"].tolist()) <= {0, 1, 2, 3}


def test_last_layer_on_the_pooled_rows_only_gives_the_full_forward_vectors(models):
    """by default the image tower's LAST LAYER runs for the pooled (first) row of every image only -- the final LayerNorm and
    the projection read nothing else: the last attention for row 0's query (all keys and values), the out-projection, fc1
    and fc2 on B rows (round 4: the MLP; round 5: attention and out-projection too).  SSW_CLIP_OPT_FULL_LAST_LAYER runs every
    row as the reference's model does.  Row 0's attention output, its f32 residual row and its bf16 copy are the full
    layer's bit for bit; what differs is the order of a few f32 sums (the row statistics are cut into six 128-column pairs
    instead of two halves, fc2's K is split over workgroups, fc1 on 200 rows takes the 128-square kernel where 10 000 rows
    take the 256-square one), and now and then a bf16 hidden value that rounds the other way.  Measured on unit vectors:
    1.1e-5 with the MLP alone (round 4), 2e-5 ... 5.0e-5 with the whole layer, over the collections of round 5; held to
    7e-5 (round 6, ADVICE r5: the measured worst case plus a margin, not twice it) -- a sixth of the tower's 4e-4 against HF, whose bar (5e-3, cos 0.999) the default form meets on its own in
    every other test of this file -- for a handful of tiles, 200 and a call that crosses the device chunk"""
    _, ours = models
    rng = np.random.default_rng(123)
    try:
        for n in (3, 200, 1030):
            tiles = rng.integers(0, 256, size=(n, 224, 224, 3), dtype=np.uint8)
            ours.set_option(ours.OPT_FULL_LAST_LAYER, False)
            pooled = ours.embed_tiles_u8(tiles, normalize=True)
            ours.set_option(ours.OPT_FULL_LAST_LAYER, True)
            full = ours.embed_tiles_u8(tiles, normalize=True)
            assert np.isfinite(pooled).all() and np.abs(pooled - full).max() <= 7e-5, (n, float(np.abs(pooled - full).max()))
    finally:
        ours.set_option(ours.OPT_FULL_LAST_LAYER, False)


def test_more_tiles_than_one_device_chunk(models):
    """1040 tiles cross the 1024-tile device chunk: row b of the result is the embedding of tile b"""
    _, ours = models
    tiles = np.random.default_rng(11).integers(0, 256, size=(1040, 224, 224, 3), dtype=np.uint8)
    whole = ours.embed_tiles_u8(tiles, normalize=True)
    assert whole.shape == (1040, 512)
    for lo, hi in ((0, 7), (250, 262), (500, 700), (1018, 1030), (1033, 1040)):
        part = ours.embed_tiles_u8(tiles[lo:hi], normalize=True)
        # a tile's embedding does not depend on its batch neighbours (same kernels, same per-row arithmetic)
        assert np.array_equal(part.view(np.uint32), whole[lo:hi].view(np.uint32)), (lo, hi)


def test_repeated_single_queries_are_bit_identical(models):
    """the same query embedded again gives the same bits (no state carried between forwards), another query of the
    same shape gives its own embedding, and a reallocation of the workspace (a larger batch in between) changes
    nothing"""
    _, ours = models
    rng = np.random.default_rng(77)

    def query(L=9):
        ids = rng.integers(0, 49405, size=(1, L)).astype(np.int32)
        ids[0, 0], ids[0, L - 1] = 49406, 49407
        return ids

    a, b = query(), query()
    first = ours.embed_text(a, normalize=True)
    second = ours.embed_text(a, normalize=True)
    third = ours.embed_text(a, normalize=True)
    assert np.array_equal(first.view(np.uint32), second.view(np.uint32))
    assert np.array_equal(first.view(np.uint32), third.view(np.uint32))
    got_b = ours.embed_text(b, normalize=True)
    both = ours.embed_text(np.concatenate([a, b]), normalize=True)
    assert np.abs(got_b - first).max() > 1e-3
    assert np.abs(both[0] - first[0]).max() < 1e-5 and np.abs(both[1] - got_b[0]).max() < 1e-5
    big = rng.integers(0, 49405, size=(300, 9)).astype(np.int32)       # forces a larger workspace
    big[:, 0], big[:, 8] = 49406, 49407
    ours.embed_text(big, normalize=True)
    again = [ours.embed_text(a, normalize=True) for _ in range(3)]
    for x in again:
        assert np.array_equal(x.view(np.uint32), first.view(np.uint32))


def test_single_query_path_agrees_with_tile_path_over_random_shapes(models):
    """every (B, L) with B x L <= 32 takes the skinny kernels (one or two 16-row tiles, one or several sequences under
    the block-causal mask, EOS anywhere): 24 random shapes against the same rows pushed through the tile kernels"""
    _, ours = models
    rng = np.random.default_rng(2024)
    shapes = [(1, 1), (1, 2), (32, 1), (16, 2), (2, 15), (1, 16), (1, 17), (4, 8)]
    while len(shapes) < 24:
        L = int(rng.integers(1, 33))
        shapes.append((int(rng.integers(1, 32 // L + 1)), L))
    for B, L in shapes:
        ids = rng.integers(0, 49405, size=(B, L)).astype(np.int32)
        ids[:, 0] = 49406
        for b in range(B):
            ids[b, int(rng.integers(0, L))] = 49407  # (a one-token row is just its end-of-text token)
        got = ours.embed_text(ids, normalize=True)
        reps = -(-33 // (B * L))
        tiled = ours.embed_text(np.tile(ids, (reps + 1, 1)), normalize=True)[:B]
        assert np.isfinite(got).all()
        assert (got * tiled).sum(1).min() >= 0.99995, (B, L)
        # (1.5e-3: the tile path folds its LayerNorms into the products since round 3 -- bf16(x) times gamma (.) W instead
        #  of bf16(LN(x)) times W -- so the two paths round differently; both are held to the HF bar above)
        assert np.abs(got - tiled).max() <= 1.5e-3, (B, L, np.abs(got - tiled).max())


# ---------------------------------------------------------------------------------------------------------------------
# round 4 (VERDICT r3 #2b, #2c): parity below and above the final embedding
# ---------------------------------------------------------------------------------------------------------------------
def _tap(lib, model, tower, layer, run):
    """residual rows behind `layer` of `tower` (lab build: ssw_clip_debug_tap) during `run()` -> f32 [rows, hidden]"""
    import ctypes
    assert lib.ssw_clip_debug_tap(model._h, tower, layer) == 0
    run()
    rows, dim = ctypes.c_int64(), ctypes.c_int32()
    assert lib.ssw_clip_debug_tap_read(model._h, None, 0, ctypes.byref(rows), ctypes.byref(dim)) == 0
    out = np.empty((rows.value, dim.value), dtype=np.float32)
    assert lib.ssw_clip_debug_tap_read(model._h, ctypes.c_void_p(out.ctypes.data), out.size, ctypes.byref(rows), ctypes.byref(dim)) == 0
    lib.ssw_clip_debug_tap(model._h, tower, -1)
    return out


# Per-layer bound: max |ours - HF| over the layer's residual rows, as a fraction of that layer's rms in HF (the rows grow
# from ~1 to ~3 over the stack), and the rms error the same way; 1.5x what the box measured (printed by the test):
#                        layer 0 .. layer 11, max|d| / rms            rms(d) / rms
#   image, bf16 rows     0.027 .. 0.089 (grows with the depth)         0.0037 .. 0.0088
#   image, f32 rows      0.011 .. 0.014 (flat)                         0.0024 .. 0.0031
#   text, tile path      0.021 .. 0.029                                0.0048 .. 0.0062
#   text, single query   0.020 .. 0.025                                0.0048 .. 0.0062
# f32 rows: the error is that of the bf16 operands of each product and does not accumulate; bf16 rows add a rounding of
# the stream itself behind every residual add (24 of them), which does.
LAYER_MAX_FRAC = {"image_bf16_rows": 0.135, "image_f32_rows": 0.022, "text_tile_f32_rows": 0.045, "text_single_query": 0.040}
LAYER_RMS_FRAC = {"image_bf16_rows": 0.0135, "image_f32_rows": 0.0046, "text_tile_f32_rows": 0.0095, "text_single_query": 0.0095}


def test_hidden_states_of_every_layer_vs_hf(lab_build):
    """VERDICT r3 #2b: the residual rows behind EVERY transformer layer of both towers against transformers'
    output_hidden_states (f32), for the image tower in both residual-row precisions, the text tower's tile path and its
    single-query path.  A kernel change that keeps the final cosine but bends a middle layer shows here."""
    import torch
    import transformers
    from seesaw_amd.models.clip import ClipModel
    lib = lab_build
    torch.manual_seed(1234)
    hf = transformers.CLIPModel(transformers.CLIPConfig()).eval()
    ours = ClipModel.from_hf(hf)  # created inside the lab build: the tap lives there
    torch.manual_seed(3)
    x = torch.randn(4, 3, 224, 224)
    rng = np.random.default_rng(8)
    ids_b = rng.integers(0, 49405, size=(3, 30)).astype(np.int64)   # 90 rows: tile path
    ids_q = rng.integers(0, 49405, size=(1, 9)).astype(np.int64)    # 9 rows: the single-query kernels
    for ids in (ids_b, ids_q):
        ids[:, 0], ids[:, -1] = 49406, 49407
    with torch.inference_mode():
        hs_img = [h.reshape(-1, 768).numpy() for h in hf.vision_model(pixel_values=x, output_hidden_states=True).hidden_states]
        hs_tb = [h.reshape(-1, 512).numpy() for h in hf.text_model(input_ids=torch.from_numpy(ids_b), output_hidden_states=True).hidden_states]
        hs_tq = [h.reshape(-1, 512).numpy() for h in hf.text_model(input_ids=torch.from_numpy(ids_q), output_hidden_states=True).hidden_states]
    cases = [("image_bf16_rows", 0, hs_img, lambda: ours.embed_image(x.numpy()), dict(image_bf16=True)),
             ("image_f32_rows", 0, hs_img, lambda: ours.embed_image(x.numpy()), dict(image_bf16=False)),
             ("text_tile_f32_rows", 1, hs_tb, lambda: ours.embed_text(ids_b.astype(np.int32)), {}),
             ("text_single_query", 1, hs_tq, lambda: ours.embed_text(ids_q.astype(np.int32)), {})]
    try:
        for name, tower, hs, run, rows in cases:
            ours.set_rows(**rows)
            worst_max = worst_rms = 0.0
            per_layer = []
            for layer in range(12):
                got = _tap(lib, ours, tower, layer, run)
                ref = hs[layer + 1]  # hidden_states[0] is the stack's input
                assert got.shape == ref.shape, (name, layer, got.shape, ref.shape)
                rms = float(np.sqrt((ref * ref).mean()))
                d = np.abs(got - ref)
                fmax, frms = float(d.max()) / rms, float(np.sqrt((d * d).mean())) / rms
                worst_max, worst_rms = max(worst_max, fmax), max(worst_rms, frms)
                per_layer.append((round(fmax, 4), round(frms, 5)))
            print(f"{name}: worst layer max|d|/rms {worst_max:.4f}, rms(d)/rms {worst_rms:.5f}; per layer {per_layer}")
            assert worst_max <= LAYER_MAX_FRAC[name] and worst_rms <= LAYER_RMS_FRAC[name], (name, worst_max, worst_rms)
    finally:
        ours.set_rows()
        ours.close()


def _synthetic_tiles(n_images, tiles_per_image, seed):
    """uint8 HWC tiles with structure: each tile a coarse random colour grid (4 .. 14 cells a side) plus a shared noise
    layer rolled by the tile number -- images differ from one another far more than white noise tiles would"""
    rng = np.random.default_rng(seed)
    n = n_images * tiles_per_image
    noise = rng.integers(-20, 21, size=(224, 224, 3), dtype=np.int16)
    out = np.empty((n, 224, 224, 3), dtype=np.uint8)
    for t in range(n):
        g = int(rng.integers(4, 15))
        cell = -(-224 // g)
        grid = rng.integers(0, 256, size=(g, g, 3), dtype=np.int16)
        big = np.repeat(np.repeat(grid, cell, axis=0), cell, axis=1)[:224, :224]
        out[t] = np.clip(big + np.roll(noise, t % 224, axis=0), 0, 255).astype(np.uint8)
    return out


def test_retrieval_level_parity_of_both_residual_forms(models):
    """VERDICT r3 #2c: an index of C5's shape (1 109 images x 13 tiles, uint8) embedded by the HIP image tower in both
    residual-row forms and by transformers in f32 (torch on the GPU, TF32 off: the f32 oracle of a floating-point
    kernel), scanned with the same HF text vector: the score error each form adds on top of the bit-exact scan, and what
    it does to the top-50 image set.  Numbers are printed and recorded in DESIGN section 4."""
    import torch
    hf, ours = models
    n_images, tpi = 1109, 13
    tiles = _synthetic_tiles(n_images, tpi, seed=5)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda", 0)
    import copy
    hf_dev = copy.deepcopy(hf).to(dev)
    ref = np.empty((tiles.shape[0], 512), dtype=np.float32)
    with torch.inference_mode():
        for lo in range(0, tiles.shape[0], 512):
            t = torch.from_numpy(tiles[lo:lo + 512]).to(dev).to(torch.float32)
            px = ((t / 255.0 - torch.from_numpy(mean).to(dev)) / torch.from_numpy(std).to(dev)).permute(0, 3, 1, 2).contiguous()
            f = hf_dev.get_image_features(pixel_values=px)
            f = f.pooler_output if hasattr(f, "pooler_output") else f
            ref[lo:lo + 512] = torch.nn.functional.normalize(f, dim=1).cpu().numpy()
        ids = np.array([[49406, 320, 1125, 539, 320, 2368, 49407]], dtype=np.int64)
        q = hf.get_text_features(input_ids=torch.from_numpy(ids))
        q = _unit((q.pooler_output if hasattr(q, "pooler_output") else q).numpy())[0].astype(np.float32)
    del hf_dev
    got = {}
    try:
        for name, f32_rows in (("bf16_rows", False), ("f32_rows", True)):
            ours.set_rows(image_bf16=not f32_rows)
            got[name] = ours.embed_tiles_u8(tiles, normalize=True)
    finally:
        ours.set_rows()

    def top50(E):
        s = (E @ q).reshape(n_images, tpi).max(1)
        return set(np.argsort(-s, kind="stable")[:50].tolist()), s

    want, s_ref = top50(ref)
    kth_gap = float(np.sort(s_ref)[::-1][49] - np.sort(s_ref)[::-1][50])
    report = {}
    for name, E in got.items():
        have, s = top50(E)
        report[name] = dict(cos_min=float((E * ref).sum(1).min()), max_tile_score_delta=float(np.abs(E @ q - ref @ q).max()),
                            max_image_score_delta=float(np.abs(s - s_ref).max()), top50_overlap=len(have & want))
    print(f"retrieval parity vs HF f32 (score spread {float(s_ref.std()):.4f}, gap at rank 50 {kth_gap:.2e}): {report}")
    for name, r in report.items():
        assert r["cos_min"] >= COS_MIN, (name, r)
        assert r["max_tile_score_delta"] <= 2.5e-3, (name, r)       # |delta E| <= 5e-3 per component x a unit query, in practice 10x less
        assert r["top50_overlap"] >= 46, (name, r)                    # what moves sits inside the score error of rank 50's neighbours
    # Measured (round 4): f32 rows 49 / 50 and |score delta| 4.0e-4, bf16 rows 48 / 50 and 1.5e-3, on scores whose spread is
    # 6.1e-3 with 1.4e-4 between ranks 50 and 51.  bf16 rows move the set more than f32 rows do, so f32 rows are the
    # default (VERDICT r3 #2c) and bf16 rows the option; the f32 form is held to the tighter figures.
    assert report["f32_rows"]["max_tile_score_delta"] <= 8e-4 and report["f32_rows"]["top50_overlap"] >= 48, report
    assert report["f32_rows"]["max_tile_score_delta"] < report["bf16_rows"]["max_tile_score_delta"], report


def test_trained_like_residual_distribution_both_forms(lab_build):
    """a random-init CLIP has none of the outlier channels / large mean-to-std ratios of a trained residual
    stream, which is where bf16 rows and the folded LayerNorm's `x W' - mean c1` cancellation lose the most.  Here the
    pre-LayerNorm of the image tower (and the token embedding of the text tower) puts values of magnitude 60-300 and a
    non-zero row mean into a few channels of every residual row; both row precisions must still meet the bar.
    Round 5: 56 tiles (four 13-tile image pyramids and a ragged rest -- 2 800 token rows: 22 row tiles of the
    128-row kernels, a ragged last one) and 16 sequences instead of 5 and 4, and the same 56 tiles once more as the tail of a
    1 056-tile call, where they straddle the 1024-tile device chunk: the vectors must be the bytes of the short call."""
    import torch
    import transformers
    from seesaw_amd.models.clip import ClipModel
    torch.manual_seed(99)
    hf = transformers.CLIPModel(transformers.CLIPConfig()).eval()
    with torch.no_grad():
        pre = hf.vision_model.pre_layrnorm
        for ch, (g, b) in {5: (30.0, 150.0), 100: (8.0, -60.0), 391: (50.0, 300.0), 767: (1.0, 90.0)}.items():
            pre.weight[ch] = g
            pre.bias[ch] = b
        emb = hf.text_model.embeddings.position_embedding.weight
        emb[:, 7] += 120.0
        emb[:, 300] -= 70.0
        emb[:, 511] += 250.0
    ours = ClipModel.from_hf(hf)
    torch.manual_seed(1)
    x = torch.randn(56, 3, 224, 224)
    rng = np.random.default_rng(12)
    ids = rng.integers(0, 49405, size=(16, 33)).astype(np.int64)
    ids[:, 0], ids[:, -1] = 49406, 49407
    with torch.inference_mode():
        hs = hf.vision_model(pixel_values=x, output_hidden_states=True).hidden_states
        r0 = hs[1].reshape(-1, 768)
        assert float(r0.abs().max()) > 150 and float((r0.mean(1).abs() / r0.std(1)).max()) > 0.02  # the stream is what the test is for
        ref_i = hf.get_image_features(pixel_values=x)
        ref_i = _unit((ref_i.pooler_output if hasattr(ref_i, "pooler_output") else ref_i).numpy())
        ref_t = hf.get_text_features(input_ids=torch.from_numpy(ids))
        ref_t = _unit((ref_t.pooler_output if hasattr(ref_t, "pooler_output") else ref_t).numpy())
    try:
        for flags in (0, 3):
            ours.set_rows(image_bf16=bool(flags & 1), text_bf16=bool(flags & 2))
            gi = ours.embed_image(x.numpy(), normalize=True)
            gt = _unit(ours.embed_text(ids.astype(np.int32), normalize=False))
            ci, ct = float((gi * ref_i).sum(1).min()), float((gt * ref_t).sum(1).min())
            print(f"outlier-channel stream, rows {'bf16/bf16' if flags else 'f32/f32'} (image/text): cos min {ci:.6f} / {ct:.6f}, "
                  f"|d| max {np.abs(gi - ref_i).max():.2e} / {np.abs(gt - ref_t).max():.2e}")
            assert ci >= COS_MIN and ct >= COS_MIN, (flags, ci, ct)
            assert np.abs(gi - ref_i).max() <= ABS_MAX and np.abs(gt - ref_t).max() <= ABS_MAX, flags
            if flags == 0:  # across the device chunk boundary (tiles 1000 .. 1055 of one call)
                big = np.concatenate([np.broadcast_to(x.numpy()[:1], (1000, 3, 224, 224)), x.numpy()])
                gb = ours.embed_image(big, normalize=True)
                del big
                assert np.array_equal(gb[1000:], gi) and np.array_equal(gb[999], gi[0])
    finally:
        ours.close()


def test_u8_tile_path_on_the_reference_normalised_tensors(models):
    """row f-3: the tiles of the tiling fixture's first image (tests/golden/tiling.npz, cut by OUR tiler, CRC-equal to the
    reference's) through ssw_clip_embed_tiles_u8, against the REFERENCE's batch_tx output for two of them (stored in
    full in the fixture) through ssw_clip_embed_image: the fused `/255, -mean, /std` of the patch gather gives the same
    bf16 patches, hence the same embedding bits."""
    import os
    import PIL.Image
    from oracle import seesaw_oracle as orc
    from seesaw_amd.indices.multiscale.multiscale_tools import generate_multiscale_tiling
    _, ours = models
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiling.npz"))
    w, h = orc.TILING_SIZES[0]
    d = generate_multiscale_tiling(PIL.Image.fromarray(orc.tiling_image(w, h, seed=100)), factor=0.5, tile_size=224, min_tile_size=224)
    tiles = np.stack(d.tile.values)[[0, 12]]
    a = ours.embed_tiles_u8(tiles, normalize=True)
    b = ours.embed_image(np.ascontiguousarray(g["im0_norm_tiles_0_12"]), normalize=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_image_embedding_with_sliding_window(models):
    """ImageEmbedding(add_slide=True) (seesaw/models/model.py:78-82): every half-overlapping 224 x 224 window of one image,
    as [1, 512, len(iis), len(jjs)]; window (i, j)'s vector is the embedding of that crop"""
    from seesaw_amd.models.embeddings import ImageEmbedding
    _, ours = models
    x = np.random.default_rng(2).standard_normal((1, 3, 448, 560)).astype(np.float32)
    out = ImageEmbedding(model=ours, add_slide=True)(preprocessed_image=x)
    assert out.shape == (1, 512, 3, 4)
    for (i, j) in ((0, 0), (1, 2), (2, 3)):
        crop = np.ascontiguousarray(x[:, :, 112 * i:112 * i + 224, 112 * j:112 * j + 224])
        assert np.array_equal(out[0, :, i, j].view(np.uint32), ours.embed_image(crop, normalize=True)[0].view(np.uint32))
