"""Parameter / result records of the session and bench layers.

Same names and fields as the reference's pydantic-v1 models (seesaw/basic_types.py:5-130)
so that configs and `summary.json` files written by either side stay interchangeable;
expressed on pydantic v2 with the v1 spellings (`.dict()`, `.copy()`, `.json()`) kept as
thin aliases because the reference's callers use them (seesaw_bench.py:238-240, 397-440).
"""
from __future__ import annotations

import importlib
from typing import List, Literal, Optional

from pydantic import BaseModel, ConfigDict


class _Record(BaseModel):
    model_config = ConfigDict(extra="allow", arbitrary_types_allowed=True)

    # pydantic-v1 spellings used throughout the reference
    def dict(self, **kw):  # noqa: A003
        return self.model_dump(**kw)

    def json(self, **kw):
        return self.model_dump_json(**kw)

    def copy(self, **kw):  # noqa: A003
        return self.model_copy(**kw)


class Box(_Record):
    """axis-aligned box in image pixels; as user feedback it carries the query text it answers and whether the
    user accepted it"""
    x1: float
    y1: float
    x2: float
    y2: float
    description: Optional[str] = None
    marked_accepted: bool = False


class Annotation(_Record):
    """a box with its own description / acceptance (textual-feedback experiments)"""
    box: Box
    description: Optional[str] = None
    marked_accepted: bool = False


class ActivationData(_Record):
    """why an image was returned: the tile that scored and its score"""
    box: Box
    score: float


class Interval(_Record):
    """time span (ms) an image spent on screen, recorded by the UI"""
    start_ms: int
    end_ms: int


class Imdata(_Record):
    """one result image as the UI / the simulated user sees it"""
    url: str
    dbidx: int
    boxes: Optional[List[Box]] = None  # None: not labelled yet; []: seen, nothing marked
    activations: Optional[List[ActivationData]] = None
    timing: List[Interval] = []


def is_image_accepted(imdata: Imdata) -> bool:
    return imdata.boxes is not None and any(b.marked_accepted for b in imdata.boxes)


class IndexSpec(_Record):
    """which dataset (d_name), which of its indices (i_name) and optionally which subset (c_name)"""
    d_name: str
    i_name: str
    c_name: Optional[str] = None  # ground-truth category naming an LVIS-style subset


StartPolicy = Literal["from_start", "after_first_batch", "after_first_negative", "after_first_positive",
                      "after_first_positive_and_negative", "after_first_reversal"]


class SessionParams(_Record):
    """everything that defines a search session: index, loop (`interactive` + its options), batch and shortlist
    sizes, per-image aggregation"""
    index_spec: IndexSpec
    interactive: str
    pass_ground_truth: Optional[bool] = False
    annotation_category: Optional[str] = None
    interactive_options: Optional[dict] = None
    batch_size: int
    index_options: Optional[dict] = {"use_vec_index": True}
    aug_larger: Literal["greater", "all", "adjacent"] = "all"
    agg_method: Optional[Literal["avg_score", "avg_vector", "plain_score"]] = "avg_score"
    shortlist_size: Optional[int] = None
    method_config: Optional[dict] = None
    image_vector_strategy: Optional[Literal["matched", "computed"]] = None
    other_params: Optional[dict] = None
    start_policy: Optional[StartPolicy] = "from_start"


class LogEntry(_Record):
    """one line of the session's action log"""
    logger: Literal["server", "client"]
    message: str
    time: float
    seen: int
    accepted: int
    other_fields: Optional[dict] = None


class SessionState(_Record):
    """the whole visible state of a session: parameters, every batch returned so far, timings, log"""
    params: SessionParams
    gdata: List[List[Imdata]]
    timing: List[float]
    reference_categories: List[str]
    query_string: Optional[str] = None
    action_log: List[LogEntry] = []


class BenchParams(_Record):
    """one benchmark run: target category, query string, number of batches, stopping rules, simulated-user knobs"""
    name: str
    sample_id: Optional[str] = None
    ground_truth_category: str
    qstr: str
    provide_textual_feedback: bool = False
    n_batches: int
    max_results: Optional[int] = None
    max_feedback: Optional[int] = None
    box_drop_prob: float = 0.0
    query_template: str = "a {}"


class BenchResult(_Record):
    """outcome of a benchmark run (found / seen counts, the final session state, timings)"""
    nimages: int
    ntotal: int
    session: SessionState
    run_info: dict
    total_time: float
    method_stats: Optional[dict] = None
    latencies: Optional[List[float]] = None


class BenchSummary(_Record):
    """what `summary.json` of a run directory holds"""
    bench_params: BenchParams
    session_params: SessionParams
    timestamp: str
    output_dir: Optional[str] = None
    result: Optional[BenchResult] = None


def get_constructor(dotted_name: str):
    """'pkg.module.Class' -> the class; index info.json files name their constructor this
    way (seesaw/indices/interface.py:37-45).  Reference package names resolve to ours."""
    if dotted_name.startswith("seesaw."):
        dotted_name = "seesaw_amd." + dotted_name[len("seesaw."):]
    module_name, _, attr = dotted_name.rpartition(".")
    return getattr(importlib.import_module(module_name), attr)
