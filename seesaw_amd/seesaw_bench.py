"""Benchmark driver: a simulated user answers every batch from ground-truth boxes.

Interface of seesaw/seesaw_bench.py: `fill_imdata` (:238-274), `benchmark_loop` (:278-355,
the unit the headline "feedback-loop iterations / s" counts: next -> simulated labels ->
update_state -> refine, timed per iteration in `latencies`), `BenchRunner.run_loop`
(:371-452, writes summary.json before and after), `summarize_session` / `process_dict` /
`get_all_session_summaries` (:457-563), `get_param_hash`, `compute_row_metrics`, `add_stats`
(:569-610), `get_bench_params`, `generate_benchmark_configs` (:627-684).  The Ray actor pool
(`make_bench_actors`, `parallel_run`) is cluster plumbing and is replaced by a plain loop.
"""
from __future__ import annotations

import copy
import glob
import hashlib
import json
import math
import os
import random
import string
import sys
import time
from contextlib import redirect_stderr, redirect_stdout

import numpy as np
import pandas as pd

from .basic_types import BenchParams, BenchResult, BenchSummary, Box, Imdata, SessionParams, is_image_accepted
from .bitmap import BitMap, FrozenBitMap
from .metrics import compute_metrics
from .seesaw_session import Session, make_session


def category2query(dataset_name: str, cat: str) -> str:
    """search term for a category (the reference keeps per-dataset tables,
    dataset_search_terms.py; synthetic datasets use the category name itself)."""
    return cat


# ground-truth category -> the class a user would say the results are being confused with (what
# `provide_textual_feedback` annotates as rejected boxes).  The reference hard-codes a table for ObjectNet
# (seesaw_bench.py:170-235, `objnet_dict`); it is dataset knowledge, not code: register the pairs of your dataset here.
objnet_dict: dict = {}


def _group_boxes(box_data: pd.DataFrame) -> dict:
    """dbidx -> (category array, list of box records): fill_imdata's per-image filter, done once"""
    cols = ["x1", "x2", "y1", "y2", "description"]
    out = {}
    if box_data.shape[0] == 0:
        return out
    order = np.argsort(box_data.dbidx.values, kind="stable")  # keep the frame's row order inside an image
    bd = box_data.iloc[order]
    recs = bd[cols].to_dict(orient="records")
    cats = bd.category.values
    ids = bd.dbidx.values
    starts = np.flatnonzero(np.r_[True, ids[1:] != ids[:-1]])
    ends = np.r_[starts[1:], ids.shape[0]]
    for s0, e0 in zip(starts.tolist(), ends.tolist()):
        out[int(ids[s0])] = (cats[s0:e0], recs[s0:e0])
    return out


def fill_imdata(imdata: Imdata, box_data: pd.DataFrame, b: BenchParams, _groups: dict = None) -> Imdata:
    """the simulated user: mark the ground-truth boxes of the target category.  `_groups` (from
    _group_boxes) replaces the per-call scan of box_data; the random draws are the same either way."""
    imdata = imdata.copy()
    boxes = []
    if _groups is not None:
        hit = _groups.get(int(imdata.dbidx))
        if hit is not None:
            cats, recs = hit
            sel = [(r, True) for c, r in zip(cats, recs) if c == b.ground_truth_category]
            if b.provide_textual_feedback:  # the confusion class's boxes, marked as rejected (seesaw_bench.py:246-258)
                confusion_class = objnet_dict[b.ground_truth_category]
                sel += [(r, False) for c, r in zip(cats, recs) if c == confusion_class]
            keep = np.random.rand(len(sel)) >= b.box_drop_prob
            boxes = [Box(marked_accepted=acc, **r) for (r, acc), k in zip(sel, keep) if k]
        imdata.boxes = boxes
        return imdata
    rows = box_data[box_data.dbidx.values == imdata.dbidx]
    if rows.shape[0] > 0:
        feedback = rows[rows.category == b.ground_truth_category].assign(marked_accepted=True)
        if b.provide_textual_feedback:
            negatives = rows[rows.category == objnet_dict[b.ground_truth_category]].assign(marked_accepted=False)
            feedback = pd.concat([feedback, negatives], axis=0, ignore_index=True)
        feedback = feedback[["x1", "x2", "y1", "y2", "description", "marked_accepted"]]
        keep = np.random.rand(feedback.shape[0]) >= b.box_drop_prob
        boxes = [Box(**r) for r in feedback[keep].to_dict(orient="records")]
    imdata.boxes = boxes
    return imdata


def benchmark_loop(*, session: Session, subset: FrozenBitMap, box_data: pd.DataFrame, b: BenchParams,
                   p: SessionParams):
    box_data = box_data.assign(description=box_data.category.map(
        lambda cat: b.query_template.format(category2query(p.index_spec.d_name, cat))))
    all_box_data = box_data
    box_data = box_data[box_data.category == b.ground_truth_category]
    positives = FrozenBitMap(box_data.dbidx.values)
    assert positives.intersection(subset) == positives, "index mismatch"
    max_results = len(positives) if b.max_results is None else min(len(positives), b.max_results)
    # textual feedback looks at every category's boxes of an image (seesaw_bench.py:327-330)
    groups = _group_boxes(all_box_data if b.provide_textual_feedback else box_data)
    total_results = total_seen = 0
    seen_dbidxs = BitMap()
    session.set_text(b.qstr)
    latencies = []
    for batch_num in range(1, b.n_batches + 1):
        start_time = time.time()
        print(f"iter {batch_num}")
        idxbatch = session.next()
        for idx in idxbatch:
            assert idx in subset, "returned a dbidx outside of range"
            assert idx not in seen_dbidxs, "returned a repeated dbidx"
            seen_dbidxs.add(idx)
        if len(idxbatch) == 0:
            break
        # get_state() builds fresh Imdata records every call and fill_imdata works on a copy,
        # so the reference's deepcopy of the whole state (seesaw_bench.py:325) is not needed
        if hasattr(session, "update_last_batch") and not os.environ.get("SSW_BENCH_FULL_STATE"):
            # only the batch just shown changes: same bookkeeping as get_state() + update_state() below, without
            # rebuilding every earlier batch on every round (seesaw_session.py)
            last_batch = [fill_imdata(imdata, box_data, b, _groups=groups) for imdata in session.last_batch()]
            session.update_last_batch(last_batch)
        else:
            s = session.get_state()
            last_batch = s.gdata[-1]
            for j, imdata in enumerate(last_batch):
                last_batch[j] = fill_imdata(imdata, box_data, b, _groups=groups)
            session.update_state(s)
        total_results += int(sum(is_image_accepted(im) for im in last_batch))
        total_seen += len(idxbatch)
        if total_results >= max_results:
            print(f"Found {total_results} (>= limit of {max_results}) for {b.ground_truth_category} "
                  f"after {batch_num} batches. stopping...")
            break
        if batch_num == b.n_batches:
            print(f"iter {batch_num} = {b.n_batches}. ending...")
            break
        if b.max_feedback is None or (batch_num + 1) * p.batch_size <= b.max_feedback:
            session.refine()
            latencies.append(time.time() - start_time)
    print(f"{latencies=}")
    return dict(nfound=int(total_results), nseen=int(total_seen), latencies=latencies)


class BenchRunner:
    def __init__(self, seesaw_root, results_dir, num_cpus: int = None, redirect_output=True, gdm=None):
        """seesaw_root: dataset root (or pass a ready `gdm` object with get_dataset(name))."""
        assert os.path.isdir(results_dir)
        if gdm is None:
            from .synthetic import GlobalDataManager
            gdm = GlobalDataManager(seesaw_root)
        self.gdm = gdm
        self.results_dir = results_dir
        random.seed(int(f"{time.time_ns()}{os.getpid()}"))
        self.redirect_output = redirect_output

    def ready(self):
        return True

    def run_loop(self, b: BenchParams, p: SessionParams):
        start = time.time()
        suffix = "".join(random.choice(string.ascii_lowercase) for _ in range(10))
        timestamp = time.strftime("%Y%m%d-%H%M%S")
        output_dir = f"{self.results_dir}/session_{timestamp}_{suffix}"
        os.mkdir(output_dir)
        summary = BenchSummary(bench_params=b, output_dir=output_dir, session_params=p, timestamp=timestamp, result=None)
        output_path = f"{output_dir}/summary.json"

        def body():
            try:
                json.dump(summary.dict(), open(output_path, "w"), indent=3)  # params first: a crash leaves a trace
                ret = make_session(self.gdm, p, b=b)
                ds = ret["dataset"]
                boxes, qgt = ds.load_ground_truth()
                gtseries = qgt[b.ground_truth_category]
                run_info = benchmark_loop(session=ret["session"], box_data=boxes,
                                          subset=BitMap(ds.file_meta.index.values), b=b, p=p)
                latencies = run_info.pop("latencies")
                session = ret["session"]
                summary.result = BenchResult(ntotal=int((gtseries > 0).sum()), nimages=int(gtseries.shape[0]),
                                             session=session.get_state(), run_info=run_info,
                                             method_stats=session.get_method_stats(),
                                             total_time=time.time() - start, latencies=latencies)
                json.dump(summary.dict(), open(output_path, "w"), indent=3)
            except Exception as exception:
                print(f"{exception=}", file=sys.stderr)
                raise

        if self.redirect_output:
            with open(f"{output_dir}/output.log", "w") as log, redirect_stdout(log), redirect_stderr(log):
                body()
        else:
            body()
        return output_dir


# ---- result summaries ---------------------------------------------------------------------
def summarize_session(res: BenchResult):
    dbidxs, accepted = [], []
    for batch in res.session.gdata:
        for imdata in batch:
            dbidxs.append(imdata.dbidx)
            accepted.append(is_image_accepted(imdata))
    accepted = np.array(accepted, dtype="int32")
    return dict(hit_indices=np.nonzero(accepted)[0].astype("int32"), dbidxs=np.array(dbidxs).astype("int32"),
                accepted=accepted, nseen=len(dbidxs), nimages=res.nimages, ntotal=res.ntotal,
                total_time=res.total_time, method_stats=res.method_stats, latencies=res.latencies)


def process_dict(obj, mode="benchmark"):
    assert mode in ["benchmark", "session"]
    if len(obj) != 1:
        bs = BenchSummary(**{k: v for k, v in obj.items() if k != "session_path"})
        b, s = bs.bench_params, bs.session_params
        res = dict(dataset=s.index_spec.d_name, index_name=s.index_spec.i_name, subset_name=s.index_spec.c_name,
                   category=b.ground_truth_category, variant=b.name, sample_id=b.sample_id, n_batches=b.n_batches,
                   batch_size=s.batch_size, max_results=b.max_results, session_params=s.json(), bench_params=b.json(),
                   has_result=bs.result is not None)
        if bs.result is not None:
            res.update(summarize_session(bs.result))
    else:
        res = dict(obj, has_result=False)
    res["session_path"] = obj["session_path"]
    return res


def process_single_result(result_path):
    path = result_path + "/summary.json"
    try:
        obj = json.load(open(path))
    except json.decoder.JSONDecodeError:
        obj = {}
    obj["session_path"] = path[: -len("summary.json")]
    return process_dict(obj)


def _have_parquet() -> bool:
    try:
        import pyarrow  # noqa: F401
        return True
    except ImportError:
        return False


def get_all_session_summaries(base_dir, force_recompute=False, parallel=True):
    """summary of every session under base_dir, cached as summary.parquet like the reference
    (pickle when no parquet engine is installed)."""
    parquet = _have_parquet()
    sumpath = base_dir + ("/summary.parquet" if parquet else "/summary.pkl")
    if not os.path.exists(sumpath) or force_recompute:
        rows = [process_single_result(os.path.dirname(p)) for p in
                glob.glob(base_dir + "/**/summary.json", recursive=True)]
        df = pd.DataFrame(rows)
        for col in ("method_stats",):
            if col in df:
                df[col] = df[col].map(lambda v: None if not v else json.dumps(v))
        df.to_parquet(sumpath) if parquet else df.to_pickle(sumpath)
    return pd.read_parquet(sumpath) if parquet else pd.read_pickle(sumpath)


def get_param_hash(dstr):
    d = json.loads(dstr)
    del d["index_spec"]
    if d.get("annotation_category", 0) is None:
        del d["annotation_category"]
    return hashlib.sha256(json.dumps(d, sort_keys=True).encode()).hexdigest()[:8]


def compute_row_metrics(row):
    if row.nseen != row.nseen:  # NaN: run without a result
        return None
    assert row.hit_indices is not None
    return compute_metrics(hit_indices=np.asarray(row.hit_indices).astype("int32"), nseen=int(row.nseen),
                           batch_size=int(row.batch_size), ntotal=int(row.ntotal), max_results=int(row.max_results))


def add_stats(summs):
    stats = summs[["hit_indices", "nseen", "batch_size", "ntotal", "max_results"]].apply(
        compute_row_metrics, axis="columns", result_type="expand")
    return summs.assign(**stats)


def get_bench_params(b_template, name, sample_id, dataset, category):
    qstr = b_template["query_template"].format(category2query(dataset, category))
    return BenchParams(**{**b_template, "qstr": qstr, "ground_truth_category": category, "name": name,
                          "sample_id": sample_id})


def get_session_params(s_template, config, index_meta):
    """template + variant config + dataset/index names -> SessionParams (configs.py:80-98)."""
    s = copy.deepcopy(s_template)
    for k, v in config.items():
        if k not in ("name", "sample_id", "max_samples", "index_name"):
            s[k] = v
    spec = dict(s.get("index_spec", {}))
    spec.update(index_meta)
    if "index_name" in config:
        spec["i_name"] = config["index_name"]
    s["index_spec"] = spec
    return SessionParams(**s)


def generate_benchmark_configs(gdm, datasets, base_configs, s_template, b_template,
                               max_classes_per_dataset=math.inf):
    ans = []
    avail = gdm.list_datasets()
    for ddict in datasets:
        if isinstance(ddict, dict):
            dataset_name, cats, default_c = ddict["name"], ddict.get("categories", []), ddict.get("subset", None)
        else:
            dataset_name, cats, default_c = ddict, [], None
        assert dataset_name in avail
        classes = gdm.get_dataset(dataset_name).load_eval_categories()
        cats = cats or classes
        for i, category in enumerate(cats):
            assert category in classes
            if i == max_classes_per_dataset:
                break
            for config in base_configs:
                c_name = default_c if default_c is not None else (category if dataset_name == "lvis" else None)
                s = get_session_params(s_template, config=config, index_meta=dict(d_name=dataset_name, c_name=c_name))
                b = get_bench_params(b_template, name=config["name"], sample_id=config.get("sample_id"),
                                     dataset=dataset_name, category=category)
                ans.append((b, s))
    return ans


def parallel_run(*, actors, tups):
    """sequential stand-in for the Ray actor pool of the reference (seesaw_bench.py:721-725)."""
    runner = actors[0] if isinstance(actors, (list, tuple)) else actors
    return [runner.run_loop(*t) for t in tups]
