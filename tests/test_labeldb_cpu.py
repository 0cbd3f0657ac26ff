"""CPU: host bookkeeping around the hot path -- LabelDB's change stamps (the key of the index's matched-tile cache) and
the session log's list semantics."""
import numpy as np

from seesaw_amd.basic_types import Box
from seesaw_amd.labeldb import LabelDB


def _box(x1=1.0, y1=2.0, x2=30.0, y2=40.0, acc=True):
    return Box(x1=x1, y1=y1, x2=x2, y2=y2, description="c0", marked_accepted=acc)


def test_put_with_unchanged_boxes_keeps_the_stamp():
    """Session.update_state (the web protocol) puts every image of every batch again each round; an unchanged label must
    keep its stamp or every seen image misses the matched-tile cache every round"""
    db = LabelDB()
    db.put(7, [_box()])
    db.put(9, None)
    db.put(11, [])
    s7, s9, s11 = db.stamp[7], db.stamp[9], db.stamp[11]
    db.put(7, [_box()])      # equal boxes, fresh objects
    db.put(9, None)
    db.put(11, [])
    assert (db.stamp[7], db.stamp[9], db.stamp[11]) == (s7, s9, s11)
    db.put(7, [_box(x2=31.0)])
    assert db.stamp[7] > s7
    db.put(9, [])            # "shown, not annotated" -> "seen, nothing relevant" is a change
    assert db.stamp[9] > s9
    db.put(11, [_box(acc=False)])
    assert db.stamp[11] > s11
    assert db.get(7, "binary") == 1 and db.get(9, "box") == []


class _Loop:
    def __init__(self):
        self.reversals = 0

    def set_reversals(self):
        self.reversals += 1


def _bare_session():
    """a Session without index / loops behind it: only the label bookkeeping of update_state / update_last_batch"""
    from seesaw_amd.bitmap import BitMap
    from seesaw_amd.seesaw_session import Session

    class _Q:
        label_db = LabelDB()

    s = object.__new__(Session)
    s.acc_indices, s.acc_activations = [], []
    s.seen, s.accepted = BitMap([]), BitMap([])
    s.timing, s.image_timing = [], {}
    s.q, s.loop, s._log_raw, s._last_change = _Q(), _Loop(), [], None
    return s


def _imdata(dbidx, accepted):
    from seesaw_amd.basic_types import Imdata
    return Imdata.model_construct(url="u", dbidx=dbidx, boxes=[_box(acc=True)] if accepted else [], activations=None, timing=[])


def test_reversal_state_survives_mixing_update_state_and_update_last_batch():
    """update_last_batch kept its reversal state in a list only it updated; a round taken through update_state
    (the web protocol, e.g. a user un-accepting an image) left it without the earlier verdicts, so set_reversals() fired
    late or never.  Now update_state refreshes it: a rejection seen through update_state and an acceptance seen through
    update_last_batch are a reversal, as _check_reversals() over the whole history says."""
    from seesaw_amd.basic_types import SessionState
    s = _bare_session()
    s.acc_indices.append(np.array([5]))
    state = SessionState.model_construct(action_log=[], gdata=[[_imdata(5, accepted=False)]], timing=[], reference_categories=[],
                                         params=None, query_string="q")
    s.update_state(state)                                   # round 1 through the web protocol: image 5 rejected
    assert s.loop.reversals == 0 and s._rev == [True, False]
    s.acc_indices.append(np.array([9]))
    s.update_last_batch([_imdata(9, accepted=True)])        # round 2 through the bench's short path: image 9 accepted
    assert s.loop.reversals == 1                            # rejected before accepted: a reversal
    assert s._check_reversals()
    # the log is a real list: a client's append stays
    n = len(s.action_log)
    s.action_log.append("client entry")
    assert len(s.action_log) == n + 1 and s.action_log[-1] == "client entry"
