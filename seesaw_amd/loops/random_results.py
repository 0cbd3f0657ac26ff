"""RandomResults: baseline that returns unseen images in random order."""
from .loop_base import LoopBase


class RandomResults(LoopBase):
    def set_text_vec(self, vec):
        super().set_text_vec(vec)

    def next_batch_external(self):
        return self.next_batch()

    def next_batch(self):
        res = self.q.query_random(self.params.batch_size) if hasattr(self.q, "query_random") else \
            self.q.query_stateful(vector=None, batch_size=self.params.batch_size)
        if hasattr(self.q, "query_random"):
            self.q.returned.update(res["dbidxs"])
        return res

    def refine(self, change=None):
        pass
