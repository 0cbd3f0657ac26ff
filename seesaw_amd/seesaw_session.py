"""Session: one user's search -- text query, batches of results, labels back, refine.

Interface of seesaw/seesaw_session.py:12-245 (`set_text`, `next`, `update_state`, `refine`,
`get_state`, `make_session`); control flow only, every numeric step is in the index / loop.
"""
from __future__ import annotations

import time

import numpy as np

from .basic_types import (ActivationData, BenchParams, Box, Imdata, LogEntry, SessionParams, SessionState,
                          is_image_accepted)
from .bitmap import BitMap
from .indices.interface import AccessMethod
from .labeldb import LabelDB
from .loops.registry import build_loop_from_params


class Session:
    """State kept per session: the batches returned so far (`acc_indices`, `acc_activations`), which images were
    seen / accepted, the label database of the query object (`q.label_db`, what the loops learn from), per-call
    timings and the action log.  The loop object (`self.loop`) owns the ranking state."""

    def __init__(self, gdm, dataset, hdb: AccessMethod, params: SessionParams, _y: np.ndarray = None):
        self.gdm, self.dataset, self.index, self.params = gdm, dataset, hdb, params
        self.acc_indices, self.acc_activations = [], []   # one entry per batch returned
        self.seen, self.accepted = BitMap([]), BitMap([])
        self.timing, self.image_timing = [], {}           # seconds per next(); UI intervals per image
        self.init_q = None                                 # the text of the query, once set
        self.q = hdb.new_query()
        if _y is not None:
            from .calibration import GroundTruthCalibrator
            assert self.index.vectors.shape[0] == _y.shape[0]
            self.q._calibrator = GroundTruthCalibrator(self.index.vectors, _y)
        self.label_db = LabelDB()  # optional prefill from ground truth (annotation_category)
        if params.annotation_category is not None:
            box_data, _ = dataset.load_ground_truth()
            df = box_data[box_data.category == params.annotation_category]
            if df.shape[0] == 0:
                print(f"warning, no entries found for category {params.annotation_category}")
            self.label_db.fill(df)
        self.loop = build_loop_from_params(gdm, self.q, params=params)
        self._log_raw = []  # (time, message, seen, accepted): LogEntry objects are built when somebody reads action_log
        self._last_change = None
        self._log("init")

    def get_totals(self):
        """counts shown by the UI header"""
        return dict(seen=len(self.seen), accepted=len(self.accepted))

    def get_method_stats(self):
        """whatever the loop wants recorded in the benchmark summary"""
        return self.loop.get_stats()

    def _log(self, message: str):
        self._log_raw.append((time.time(), message, len(self.seen), len(self.accepted)))

    @property
    def action_log(self):
        """the reference's list of LogEntry (seesaw_session.py:56-62), materialised on demand: five pydantic objects a
        round were a tenth of a `plain` round at LVIS-subset size"""
        raw = self._log_raw  # materialised in place and returned itself: a client's `session.action_log.append(...)` stays
        for i, e in enumerate(raw):
            if isinstance(e, tuple):
                raw[i] = LogEntry.model_construct(logger="server", time=e[0], message=e[1], seen=e[2], accepted=e[3])
        return raw

    @action_log.setter
    def action_log(self, entries):  # update_state: the client hands the log back with its own entries in it
        self._log_raw = list(entries)

    def next(self):
        """next batch of image ids from the loop; the batch and its activations are remembered for get_state"""
        self._log("next.start")
        start = time.time()
        r = self.loop.next_batch_external()
        self.timing.append(time.time() - start)
        self.acc_indices.append(r["dbidxs"])
        self.acc_activations.append(r["activations"])
        self._log("next.end")
        return r["dbidxs"]

    def set_text(self, key):
        """embed the text (CLIP text tower through the index) and hand the unit vector to the loop"""
        self._log("set_text")
        self.init_q = key
        self.loop.state.curr_str = key
        vec = self.index.string2vec(string=key)
        # as in the reference, LoopState.tvec is never assigned (seesaw_session.py:96-103 sets curr_str only),
        # so LogReg2 / PseudoLR build their scorers with regularizer_vector=None: no regulariser at all.
        # Setting it here would "repair" them and change every result of those two loops.
        self.loop.set_text_vec(vec)

    def update_state(self, state: SessionState):
        """take the client's view of the session (labels drawn on the returned images) as the new truth"""
        self._update_labeldb(state)
        self._log("update_state.end")
        reversal = self._check_reversals()
        # update_last_batch() keeps this state incrementally; a session may mix both entry points (a client un-accepting
        # an earlier image goes through here), so the full walk refreshes it
        shown = [int(i) for b in self.acc_indices for i in np.asarray(b).reshape(-1)]
        self._rev = [any(i not in self.accepted for i in shown), reversal or self._reversal_ignoring_totals(shown)]
        if reversal:
            self.loop.set_reversals()

    # ---- the last batch only ---------------------------------------------------------------------------------
    # benchmark_loop's simulated user labels the batch it was just shown and never revisits an earlier one, yet
    # get_state() / update_state() (the web client's protocol, kept above) rebuild every batch so far on every round:
    # O(rounds) records, bitmap copies and label-db writes per round.  These two do the same bookkeeping for the
    # last batch alone; every set / label / change list ends up identical (tests: the reference's session sequences).
    def last_batch(self):
        """the Imdata records of the batch next() just returned (what get_state().gdata[-1] holds)"""
        prefill = self.params.annotation_category is not None
        return self.get_panel_data(idxbatch=self.acc_indices[-1], activation_batch=self.acc_activations[-1], prefill=prefill)

    def update_last_batch(self, batch):
        """update_state() for a state in which only the last batch changed"""
        seen, accepted, put = self.seen, self.accepted, self.q.label_db.put
        change = []
        for imdata in batch:
            dbidx = imdata.dbidx
            self.image_timing[dbidx] = imdata.timing
            new_seen = dbidx not in seen
            if new_seen:
                seen.add(dbidx)
            acc = is_image_accepted(imdata)
            new_acc = acc and dbidx not in accepted
            if new_acc:
                accepted.add(dbidx)
            put(dbidx, imdata.boxes)
            if new_seen or new_acc:
                change.append((dbidx, 1 if new_acc else 0))
            # reversal = an accepted image shown after a rejected one (sticky: earlier verdicts do not change here)
            rev = self.__dict__.setdefault("_rev", [False, False])  # [a rejected image was shown, reversal seen]
            if acc and rev[0]:
                rev[1] = True
            elif not acc:
                rev[0] = True
        self._last_change = sorted(change)  # ascending dbidx, as the bitmap union of update_state() lists them
        self._log("update_state.end")
        if self.__dict__.get("_rev", [False, False])[1] and len(accepted) != 0 and len(accepted) != len(seen):
            self.loop.set_reversals()

    def _reversal_ignoring_totals(self, shown):
        """an accepted image shown after a rejected one, without _check_reversals()'s all / none shortcuts (those are
        re-applied by update_last_batch each round, on the totals of that round)"""
        rejected_before = False
        for i in shown:
            if i not in self.accepted:
                rejected_before = True
            elif rejected_before:
                return True
        return False

    def _check_reversals(self):
        """a reversal = some rejected image shown before an accepted one."""
        if len(self.accepted) == 0 or len(self.accepted) == len(self.seen):
            return False
        rejected_before = False
        for batch in self.acc_indices:
            for idx in np.asarray(batch).reshape(-1):
                if int(idx) not in self.accepted:
                    rejected_before = True
                elif rejected_before:
                    return True
        return False

    def refine(self):
        """let the loop learn from the labels that changed since the last call"""
        self._log("refine.start")
        self.loop.refine_external(self._last_change)
        self._log("refine.end")

    def get_state(self) -> SessionState:
        """every batch so far as Imdata records (boxes from the label database), plus log and timings"""
        gdata = []
        last = len(self.acc_indices) - 1
        for i, (indices, accs) in enumerate(zip(self.acc_indices, self.acc_activations)):
            prefill = (self.params.annotation_category is not None) and (i == last)
            gdata.append(self.get_panel_data(idxbatch=indices, activation_batch=accs, prefill=prefill))
        return SessionState.model_construct(action_log=self.action_log, gdata=gdata, timing=self.timing,
                                            reference_categories=[], params=self.params,
                                            query_string=self.loop.state.curr_str)

    def _static_panel(self, idxbatch, activation_batch):
        """url + activation records of one batch: they never change once returned, so they are
        converted from the index's DataFrames once (the reference redoes it for every batch on
        every get_state, seesaw_session.py:153-186)."""
        key = id(idxbatch)
        cache = self.__dict__.setdefault("_panel_cache", {})
        hit = cache.get(key)
        if hit is None:
            urls = self.dataset.get_urls(idxbatch)
            acts = []
            if hasattr(activation_batch, "records") and len(activation_batch) == len(idxbatch):
                # the index's own result object: numbers straight from its arrays, no DataFrame in between
                acts = [[ActivationData(box=Box(x1=v[0], y1=v[1], x2=v[2], y2=v[3]), score=v[4])]
                        for v in activation_batch.records()]
            else:
                for i in range(len(idxbatch)):
                    if activation_batch is None or len(activation_batch) == 0:
                        acts.append(None)
                        continue
                    vals = activation_batch[i][["x1", "y1", "x2", "y2", "score"]].to_numpy(dtype=np.float64).tolist()
                    acts.append([ActivationData(box=Box(x1=v[0], y1=v[1], x2=v[2], y2=v[3]), score=v[4]) for v in vals])
            hit = cache[key] = (idxbatch, urls, acts)  # keeps idxbatch alive so its id stays unique
        return hit[1], hit[2]

    def get_panel_data(self, *, idxbatch, activation_batch=None, prefill=False):
        urls, acts = self._static_panel(idxbatch, activation_batch)
        db = self.label_db if prefill else self.q.label_db
        out = []
        cache = self.__dict__.setdefault("_imdata_cache", {})
        for url, dbidx, activations in zip(urls, idxbatch, acts):
            dbidx = int(dbidx)
            boxes, timing = db.get(dbidx, format="box"), self.image_timing.get(dbidx, None)
            # get_state() is called every round and walks every batch so far: the record of an image whose
            # boxes / timing objects have not been replaced since the last call is reused as it is
            hit = cache.get((dbidx, prefill))
            if hit is not None and hit[0] is boxes and hit[1] is timing:
                out.append(hit[2])
                continue
            im = Imdata.model_construct(url=url, dbidx=dbidx, boxes=boxes, activations=activations,
                                        timing=[] if timing is None else timing)
            if boxes is not None:  # (None = never shown: db.get builds nothing to key on)
                cache[(dbidx, prefill)] = (boxes, timing, im)
            out.append(im)
        return out

    def _update_labeldb(self, state: SessionState):
        # rebuilt from scratch every time: the user may have un-accepted an image
        self.action_log = state.action_log
        old_accepted, old_seen = self.accepted.copy(), self.seen.copy()
        self.accepted.clear()
        self.seen.clear()
        for batch in state.gdata:
            for imdata in batch:
                self.image_timing[imdata.dbidx] = imdata.timing
                self.seen.add(imdata.dbidx)
                if is_image_accepted(imdata):
                    self.accepted.add(imdata.dbidx)
                self.q.label_db.put(imdata.dbidx, imdata.boxes)
        delta_accepted = self.accepted - old_accepted
        delta_seen = self.seen - old_seen
        self._last_change = [(idx, 1 if idx in delta_accepted else 0) for idx in delta_seen.union(delta_accepted)]


def make_session(gdm, p: SessionParams, b: BenchParams = None):
    ds = gdm.get_dataset(p.index_spec.d_name)
    if p.index_spec.c_name is not None:
        ds = ds.load_subset(p.index_spec.c_name)
    _y = None
    if p.pass_ground_truth:
        _, gt = ds.load_ground_truth()
        _y = gt[b.ground_truth_category]
    idx = ds.load_index(p.index_spec.i_name, options=p.index_options)
    session = Session(gdm, ds, idx, p, _y=_y)
    return {"session": session, "dataset": ds}
