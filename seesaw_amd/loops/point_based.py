"""Loops whose whole state is one query vector (seesaw/loops/point_based.py:3-27): `PointBased` keeps
`curr_vec` and asks the index for the next batch with it; `Plain` is the no-feedback baseline."""
from .loop_base import LoopBase


class PointBased(LoopBase):
    curr_vec = None  # set by set_text_vec, replaced by the subclasses' refine()

    def set_text_vec(self, vec):
        LoopBase.set_text_vec(self, vec)
        self.curr_vec = vec

    def next_batch(self):
        if self.curr_vec is None:
            raise AssertionError("no query vector yet: set_text_vec comes first")
        return self._next_batch_curr_vec(self.curr_vec)


class Plain(PointBased):
    def refine(self, change=None):
        """feedback is ignored: every batch is ranked by the text vector"""
        return None
