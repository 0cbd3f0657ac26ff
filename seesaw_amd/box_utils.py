"""Box geometry on the few hundred boxes a query touches (host side, numpy).

Covers what the hot path's callers need from the reference's seesaw/box_utils.py:
`box_iou` (:336-350, there via torchvision's _box_inter_union), `box_join` (:364-372) and
`left_iou_join` (:406-420).  Out of the GPU scope by SURVEY section 2 (#19): at most
shortlist_size images x tiles boxes per call.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

_XYXY = ["x1", "y1", "x2", "y2"]


def _xyxy(df) -> np.ndarray:
    """[n, 4] in the dtype np.stack gives the four columns (the reference's df2tensor, box_utils.py:326-333):
    float32 tile boxes stay float32, python-float label boxes are float64"""
    arr = np.stack([np.asarray(df[c].values) for c in _XYXY], axis=1)
    return arr if arr.dtype.kind == "f" else arr.astype(np.float32)


def box_iou(df1, df2, return_containment: bool = False):
    """pairwise IoU [len(df1), len(df2)]; containment = intersection / area(df1 box).  Dtypes follow the
    reference (torchvision's _box_inter_union on the two tensors): each side's areas in its own dtype,
    intersections and the quotient in the promoted dtype -- float32 throughout for two float32 frames."""
    a, b = _xyxy(df1), _xyxy(df2)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    w = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0])
    h = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1])
    inter = np.clip(w, 0, None) * np.clip(h, 0, None)
    union = (area_a[:, None] + area_b[None, :]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union
        cont = inter / area_a[:, None]
    return (iou, cont) if return_containment else iou


def box_join(df1: pd.DataFrame, df2: pd.DataFrame, iou_gt: float = 0.0) -> pd.DataFrame:
    """all pairs (i, j) with IoU > iou_gt, with both rows' columns suffixed _left/_right."""
    iou, cont = box_iou(df1, df2, return_containment=True)
    ii, jj = np.nonzero(iou > iou_gt)
    left = df1.iloc[ii].reset_index(drop=True).add_suffix("_left")
    right = df2.iloc[jj].reset_index(drop=True).add_suffix("_right")
    head = pd.DataFrame({"iloc_left": ii, "iloc_right": jj, "iou": iou[ii, jj], "cont": cont[ii, jj]})
    return pd.concat([head, left, right], axis=1)


def left_iou_join(vector_meta_df: pd.DataFrame, boxes: pd.DataFrame) -> pd.DataFrame:
    """for every tile row: max IoU with any label box of the same image (0 if none)."""
    if vector_meta_df.shape[0] == 0:
        return vector_meta_df.assign(max_iou=np.zeros(0))
    max_iou = np.zeros(vector_meta_df.shape[0])
    positions = np.arange(vector_meta_df.shape[0])
    dbidx = vector_meta_df["dbidx"].values
    box_dbidx = boxes["dbidx"].values if boxes.shape[0] else np.zeros(0, dtype=np.int64)
    for d in np.unique(dbidx):
        sel = box_dbidx == d
        if not sel.any():
            continue
        rows = positions[dbidx == d]
        iou = box_iou(vector_meta_df.iloc[rows], boxes[sel])
        iou = np.where(iou > 0, iou, 0.0)
        max_iou[rows] = iou.max(axis=1)
    return vector_meta_df.assign(max_iou=max_iou)
