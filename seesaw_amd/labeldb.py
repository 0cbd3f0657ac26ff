"""Per-session store of the box labels the user (or the simulated user) has given.

Interface of the reference's LabelDB (seesaw/labeldb.py:5-75): `put(dbidx, boxes)`,
`get(dbidx, format)`, `get_seen()`, `get_box_df()`, `fill(df)`.  `boxes` is None for
"shown but not annotated", [] for "seen, nothing relevant".
"""
from __future__ import annotations

from typing import Dict, List, Optional

import pandas as pd

from .basic_types import Box
from .bitmap import BitMap

_BOX_COLS = ["x1", "x2", "y1", "y2"]


class LabelDB:
    def __init__(self):
        self.ldata: Dict[int, Optional[List[Box]]] = {}
        self.stamp: Dict[int, int] = {}  # dbidx -> serial number of its latest put(): a cheap "did this image's labels change"
        self._serial = 0
        self.changes: List[int] = []     # the dbidx of every put() that changed something, in order (entry i has serial i + 1)

    def put(self, dbidx: int, boxes: Optional[List[Box]]):
        dbidx = int(dbidx)
        # the web protocol (Session.update_state) writes every image of every batch again each round: an unchanged
        # label keeps its stamp, so whatever is cached per (image, stamp) -- the index's matched-tile table -- still hits
        unchanged = dbidx in self.ldata and self.ldata[dbidx] == boxes and (self.ldata[dbidx] is None) == (boxes is None)
        self.ldata[dbidx] = boxes
        if not unchanged:
            self._serial += 1
            self.stamp[dbidx] = self._serial
            self.changes.append(dbidx)

    def get_seen(self) -> BitMap:
        return BitMap(self.ldata.keys())

    def fill(self, df: pd.DataFrame):
        """prefill from a ground-truth box table (columns x1,y1,x2,y2,category,dbidx)."""
        for dbidx, rows in df.groupby("dbidx"):
            boxes = [Box(x1=r.x1, y1=r.y1, x2=r.x2, y2=r.y2, description=r.category, marked_accepted=True)
                     for r in rows.itertuples()]
            self.put(int(dbidx), boxes)

    def get(self, dbidx: int, format: str):  # noqa: A002
        dbidx = int(dbidx)
        if dbidx not in self.ldata:
            return None  # never shown
        boxes = self.ldata[dbidx]
        if boxes is None:
            boxes = []
        if format == "box":
            return boxes
        if format == "binary":
            return int(len(boxes) > 0)
        if format == "df":
            recs = [[getattr(b, c) for c in _BOX_COLS] for b in boxes]
            return pd.DataFrame(recs, columns=_BOX_COLS).astype("float32")
        raise AssertionError(f"unknown format {format}")

    def get_box_df(self, return_description: bool = False) -> pd.DataFrame:
        cols = ["dbidx"] + (["description", "marked_accepted"] if return_description else []) + _BOX_COLS
        recs = []
        for dbidx, boxes in self.ldata.items():
            for b in boxes or []:
                rec = {"dbidx": dbidx, "x1": b.x1, "x2": b.x2, "y1": b.y1, "y2": b.y2}
                if return_description:
                    rec["description"] = b.description
                    rec["marked_accepted"] = b.marked_accepted
                recs.append(rec)
        df = pd.DataFrame(recs, columns=cols)
        df = df.astype({c: "float32" for c in _BOX_COLS})
        df["dbidx"] = df["dbidx"].astype("int32")
        if not return_description:
            return df
        if df.shape[0] == 0:  # the reference's empty frame carries these two columns too
            df["description"] = df["description"].astype("float32")
            df["marked_accepted"] = df["marked_accepted"].astype("float32")
        return df
