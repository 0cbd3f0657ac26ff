"""CPU: host bookkeeping around the hot path -- LabelDB's change stamps (the key of the index's matched-tile cache) and
the session log's list semantics."""
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
