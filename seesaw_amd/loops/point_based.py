"""Loops whose state is a single query vector (seesaw/loops/point_based.py:3-27)."""
from .loop_base import LoopBase


class PointBased(LoopBase):
    def __init__(self, gdm, q, params):
        super().__init__(gdm, q, params)
        self.curr_vec = None

    def set_text_vec(self, vec):
        super().set_text_vec(vec)
        self.curr_vec = vec

    def next_batch(self):
        assert self.curr_vec is not None
        return self._next_batch_curr_vec(self.curr_vec)


class Plain(PointBased):
    """no feedback: keep querying with the text vector."""

    @staticmethod
    def from_params(gdm, q, params):
        return Plain(gdm, q, params)

    def refine(self, change=None):
        pass
