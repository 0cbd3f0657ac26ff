"""Offline multiscale tiling and index construction: the reference's interface
(seesaw/indices/multiscale/multiscale_tools.py:10-261) without Ray.

Host side (PIL / numpy, as in the reference): image pyramid, strided 224x224 tiling and the per-tile
metadata.  GPU side: tiles go to the device as uint8 HWC and `batch_tx`'s `/255 -> normalise` is fused
into CLIP's patch gather (ssw_clip_embed_tiles_u8), so what the reference runs as
`map_batches(batch_tx) -> map_batches(InferenceActor)` (:205-221) is one call per 200 tiles.
On disk the index keeps the reference's layout: `<dataset>/indices/<name>/info.json` and
`vectors.sorted.cached` (parquet: dbidx, file_path, patch_id, zoom_level, x1, y1, x2, y2, scale_factor,
max_zoom_level, vectors), rows sorted by dbidx.
"""
from __future__ import annotations

import io
import json
import math
import os
import warnings

import numpy as np
import pandas as pd
import PIL
import PIL.Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def rescale(im, scale, min_size):
    (w, h) = im.size
    target_w = max(math.floor(w * scale), min_size)
    target_h = max(math.floor(h * scale), min_size)
    return im.resize(size=(target_w, target_h), resample=PIL.Image.BILINEAR)


def pyramid(im, factor, abs_min):
    """image pyramid from the smallest side == abs_min up to (at least) the original size, one level per
    1/factor; returned smallest scale factor first (multiscale_tools.py:16-49)."""
    assert factor < 1.0
    factor = 1.0 / factor
    size = min(im.size)
    start_size = max(size, abs_min)
    start_scale = start_size / size
    end_scale = abs_min / size
    ntimes = math.ceil(math.log(start_scale / end_scale) / math.log(factor))
    start_size = math.ceil(math.exp(ntimes * math.log(factor) + math.log(abs_min)))
    start_scale = start_size / size
    factors = np.geomspace(start=start_scale, stop=end_scale, num=ntimes + 1, endpoint=True).tolist()
    ims = [rescale(im, scale=sf, min_size=abs_min) for sf in factors]
    assert len(ims) > 0
    assert min(ims[0].size) >= abs_min
    assert min(ims[-1].size) == abs_min
    df = pd.DataFrame({"image": ims, "scale_factor": factors, "zoom_level": np.arange(len(ims))})
    return df.sort_values("scale_factor", ascending=True).reset_index(drop=True)


def rearrange_into_tiles(img1, tile_size):
    """non-overlapping tiles of the top-left (h // t) x (w // t) grid, row-major, plus their boxes"""
    arr = np.asarray(img1)
    (h, w, c) = arr.shape
    nh, nw = h // tile_size, w // tile_size
    arr = arr[: nh * tile_size, : nw * tile_size]
    patches = arr.reshape(nh, tile_size, nw, tile_size, c).transpose(0, 2, 1, 3, 4).reshape(nh * nw, tile_size, tile_size, c)
    ii, jj = np.meshgrid(np.arange(nh).astype("int32"), np.arange(nw).astype("int32"), indexing="ij")
    x1 = jj.reshape(-1) * tile_size
    y1 = ii.reshape(-1) * tile_size
    return pd.DataFrame({"tile": list(np.ascontiguousarray(patches)), "x1": x1, "y1": y1, "x2": x1 + tile_size,
                         "y2": y1 + tile_size})


def _process_shifted(base_arr, tile_size, shift_y, shift_x):
    res = rearrange_into_tiles(base_arr[shift_y:, shift_x:], tile_size=tile_size)
    res["y1"] = res["y1"] + shift_y
    res["y2"] = res["y2"] + shift_y
    res["x1"] = res["x1"] + shift_x
    res["x2"] = res["x2"] + shift_x
    return res


def strided_tiling(img1, tile_size):
    """tiles at stride tile_size / 2: the four half-tile shifts of the plain grid (:85-95)"""
    base_arr = np.asarray(img1)
    stride_size = tile_size // 2
    all_res = []
    for i in [0, 1]:
        for j in [0, 1]:
            all_res.append(_process_shifted(base_arr, shift_x=stride_size * i, shift_y=stride_size * j, tile_size=tile_size))
    return pd.concat(all_res, ignore_index=True)


def generate_multiscale_tiling(im, tile_size, factor, min_tile_size):
    pdf = pyramid(im, factor=factor, abs_min=tile_size)
    mask = (224 / pdf.scale_factor >= min_tile_size) | (pdf.index == 0)  # keep the coarsest level at least
    pdf = pdf[mask]
    assert pdf.shape[0] > 0, "mask eliminated all images. should keep at least one"
    acc = []
    max_zoom_level = pdf.zoom_level.max()
    for tup in pdf.itertuples():
        df = strided_tiling(tup.image, tile_size=tile_size)
        df = df.assign(scale_factor=np.float32(tup.scale_factor), zoom_level=np.int16(tup.zoom_level))
        df = df.assign(**(df[["x1", "x2", "y1", "y2"]] / tup.scale_factor).astype("float32"))
        acc.append(df)
    batch_df = pd.concat(acc, ignore_index=True)
    batch_df = batch_df.assign(patch_id=np.arange(batch_df.shape[0], dtype=np.int16),
                               max_zoom_level=np.int16(max_zoom_level))
    return batch_df


def reconstruct_patch(im1, meta_tup):
    """the reference's test helper, expression for expression (multiscale_tools.py:124-128).  It divides by
    the scale factor where the tiler above also divided, so it only inverts the tiler at scale factor 1
    (which is all the reference's own strided-tiling test exercises); tile_source_box is the exact inverse."""
    sf = meta_tup.get("scale_factor", 1)
    adjusted_im = rescale(im1, scale=sf, min_size=224)
    return adjusted_im.crop((math.ceil(meta_tup.x1 / sf), math.ceil(meta_tup.y1 / sf), math.ceil(meta_tup.x2 / sf),
                             math.ceil(meta_tup.y2 / sf)))


def tile_source_box(meta_tup):
    """pixel box of a tile inside its own pyramid level (box in original-image coordinates x scale factor)"""
    sf = float(meta_tup.scale_factor)
    return tuple(int(round(float(v) * sf)) for v in (meta_tup.x1, meta_tup.y1, meta_tup.x2, meta_tup.y2))


def multiscale_preproc_tup(rowtup, min_tile_size):
    try:
        image = PIL.Image.open(io.BytesIO(rowtup.bytes)).convert("RGB")
        tile_df = generate_multiscale_tiling(image, factor=0.5, tile_size=224, min_tile_size=min_tile_size)
    except PIL.UnidentifiedImageError:
        warnings.warn(f"error parsing binary {rowtup.file_path}. Ignoring...")
        tile_df = None
    return tile_df


def multiscale_preproc_batch(batch_df, min_tile_size):
    dfs = []
    for tup in batch_df.itertuples():
        tile_df = multiscale_preproc_tup(tup, min_tile_size=min_tile_size)
        if tile_df is None:
            continue
        tile_df = tile_df.assign(**{k: v for k, v in tup._asdict().items() if k not in ("bytes", "Index")})
        dfs.append(tile_df)
    res = pd.concat(dfs, ignore_index=True)
    for i, c in enumerate(["dbidx", "file_path", "patch_id"]):
        colval = res[c]
        res = res.drop([c], axis=1)
        res.insert(i, c, colval)
    return res


def batch_tx(batch_df):
    """uint8 HWC tiles -> normalised f32 CHW (multiscale_tools.py:167-183).  Host restatement kept for
    API parity and tests; the indexing pipeline below never calls it (the GPU does this inside
    ssw_clip_embed_tiles_u8)."""
    arr = np.stack(batch_df.tile.values).astype(np.float32)
    tmp01 = np.transpose(arr, (0, 3, 1, 2)) / np.float32(255.0)
    mean = np.asarray(CLIP_MEAN, dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(CLIP_STD, dtype=np.float32).reshape(1, 3, 1, 1)
    return batch_df.assign(tile=list(((tmp01 - mean) / std).astype(np.float32)))


class InferenceActor:
    """tiles -> CLIP image vectors (multiscale_tools.py:187-202: 200 tiles per call there).  A tile's vector does not
    depend on the tiles it shares a call with, so the call size is a throughput choice: 1024 here (10.4 us a tile on an
    MI355X against 12.2 at 200)."""

    def __init__(self, model, batch_size: int = 1024):
        self.model = model  # seesaw_amd.models.clip.ClipModel (or anything with embed_tiles_u8)
        self.batch_size = batch_size

    def __call__(self, batch_df):
        tiles = batch_df.tile.values
        out = []
        for s in range(0, len(tiles), self.batch_size):
            out.append(self.model.embed_tiles_u8(np.stack(tiles[s:s + self.batch_size]), normalize=True))
        vecs = np.concatenate(out) if out else np.zeros((0, 512), np.float32)
        return batch_df.drop(["tile"], axis=1).assign(vectors=list(vecs))


def run_multiscale_extraction_pipeline(image_rows: pd.DataFrame, model, vector_output_path, min_tile_size,
                                       images_per_batch: int = 16):
    """image_rows: DataFrame(dbidx, file_path, bytes).  Tiles `images_per_batch` images at a time on the
    host, embeds on the GPU, writes one parquet file with rows sorted by dbidx."""
    actor = InferenceActor(model)
    parts = []
    image_rows = image_rows.sort_values("dbidx").reset_index(drop=True)
    for s in range(0, image_rows.shape[0], images_per_batch):
        tiles = multiscale_preproc_batch(image_rows.iloc[s:s + images_per_batch], min_tile_size=min_tile_size)
        parts.append(actor(tiles))
    df = pd.concat(parts, ignore_index=True)
    write_vector_parquet(df, vector_output_path)
    return df


def write_vector_parquet(df: pd.DataFrame, path: str):
    import pyarrow as pa
    import pyarrow.parquet as pq
    vecs = np.stack(df.vectors.values).astype(np.float32)
    cols = {c: pa.array(df[c].values) for c in df.columns if c != "vectors"}
    cols["vectors"] = pa.FixedSizeListArray.from_arrays(pa.array(vecs.reshape(-1)), vecs.shape[1])
    os.makedirs(path, exist_ok=True)
    pq.write_table(pa.table(cols), os.path.join(path, "part-0.parquet"))


def read_vector_parquet(path: str):
    """-> (meta DataFrame without the vector column, vectors f32 [N, d])"""
    import pyarrow.parquet as pq
    table = pq.read_table(path)
    col = table.column("vectors").combine_chunks()
    d = col.type.list_size
    vecs = np.asarray(col.flatten().to_numpy(zero_copy_only=False), dtype=np.float32).reshape(-1, d)
    return table.drop(["vectors"]).to_pandas().reset_index(drop=True), vecs


def create_multiscale_index(*, image_rows: pd.DataFrame, dataset_path: str, index_name: str, model, model_path: str,
                            min_tile_size=224, force=False):
    """multiscale_tools.py:225-261: writes <dataset>/indices/<name>/{info.json, vectors.sorted.cached} and
    returns the index path.  `model` embeds the tiles, `model_path` is what info.json records (it is what
    MultiscaleIndex.from_path hands to load_embedding for the text side)."""
    index_output_path = f"{dataset_path}/indices/{index_name}"
    if os.path.exists(index_output_path) and not force:
        raise FileExistsError(index_output_path)
    tmp = index_output_path + ".tmp"
    os.makedirs(tmp, exist_ok=True)
    info = {"constructor": "seesaw.indices.multiscale.multiscale_index.MultiscaleIndex", "model": model_path,
            "dataset": os.path.abspath(dataset_path)}
    with open(f"{tmp}/info.json", "w") as f:
        json.dump(info, f, indent=2)
    run_multiscale_extraction_pipeline(image_rows, model, f"{tmp}/vectors.sorted.cached", min_tile_size=min_tile_size)
    if os.path.exists(index_output_path):
        import shutil
        shutil.rmtree(index_output_path)
    os.replace(tmp, index_output_path)
    return index_output_path
