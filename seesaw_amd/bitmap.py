"""Integer sets with the slice of the pyroaring API the seesaw interfaces use.

The reference passes `pyroaring.BitMap` / `FrozenBitMap` objects through its public
interface (`exclude=` of AccessMethod.query, InteractiveQuery.returned,
AccessMethod.all_indices, benchmark_loop's `subset`; seesaw/query_interface.py:20,
seesaw/indices/interface.py:16-25).  When pyroaring is installed those very classes are
re-exported; otherwise these sorted-numpy stand-ins provide the same behaviour for the
operations the hot path relies on: ascending iteration (`np.array(bm)` is sorted --
coarse_index.py:76 depends on it), `rank`, set algebra, membership, `update`.
Host-side bookkeeping only: the device gets the ids as a plain int64 list
(ssw_index_set_excluded) and keeps its own bitmap in HBM.
"""
from __future__ import annotations

from typing import Iterable

import numpy as np

try:  # pragma: no cover - not installed in the build image
    from pyroaring import BitMap, FrozenBitMap  # type: ignore
    HAVE_PYROARING = True
except ImportError:
    HAVE_PYROARING = False

    def _as_sorted(values) -> np.ndarray:
        if values is None:
            return np.zeros(0, dtype=np.int64)
        if isinstance(values, _IntSet):
            return values._v
        arr = np.fromiter((int(v) for v in values), dtype=np.int64) if not isinstance(values, np.ndarray) \
            else values.astype(np.int64, copy=False).reshape(-1)
        return np.unique(arr)

    class _IntSet:
        __slots__ = ("_v",)

        def __init__(self, values: Iterable[int] = None):
            self._v = _as_sorted(values)

        # ---- queries
        def __len__(self):
            return int(self._v.shape[0])

        def __iter__(self):
            return iter(self._v.tolist())

        def __contains__(self, x):
            x = int(x)
            i = np.searchsorted(self._v, x)
            return bool(i < self._v.shape[0] and self._v[i] == x)

        def __array__(self, dtype=None, copy=None):
            return self._v if dtype is None else self._v.astype(dtype)

        def __getitem__(self, i):
            return int(self._v[i])

        def rank(self, x) -> int:
            """number of members <= x (pyroaring semantics; coarse_index.py:117)."""
            return int(np.searchsorted(self._v, int(x), side="right"))

        def min(self):
            return int(self._v[0])

        def max(self):
            return int(self._v[-1])

        def to_array(self) -> np.ndarray:
            return self._v.copy()

        def __eq__(self, other):
            return isinstance(other, _IntSet) and np.array_equal(self._v, other._v)

        def __hash__(self):
            return hash(self._v.tobytes())

        def __repr__(self):
            head = ", ".join(str(x) for x in self._v[:6].tolist())
            return f"{type(self).__name__}([{head}{', ...' if len(self) > 6 else ''}])"

        # ---- algebra (results keep the left operand's type)
        def _new(self, arr):
            out = type(self).__new__(type(self))
            out._v = arr
            return out

        def union(self, *others):
            arrs = [self._v] + [_as_sorted(o) for o in others]
            return self._new(np.unique(np.concatenate(arrs)))

        def intersection(self, *others):
            v = self._v
            for o in others:
                v = np.intersect1d(v, _as_sorted(o), assume_unique=True)
            return self._new(v)

        def difference(self, *others):
            v = self._v
            for o in others:
                v = np.setdiff1d(v, _as_sorted(o), assume_unique=True)
            return self._new(v)

        def intersection_cardinality(self, other) -> int:
            return int(np.intersect1d(self._v, _as_sorted(other), assume_unique=True).shape[0])

        def difference_cardinality(self, other) -> int:
            return len(self) - self.intersection_cardinality(other)

        def issubset(self, other) -> bool:
            return self.intersection_cardinality(other) == len(self)

        __or__ = union
        __and__ = intersection
        __sub__ = difference

        def copy(self):
            return self._new(self._v.copy())

    class FrozenBitMap(_IntSet):
        """immutable"""

    class BitMap(_IntSet):
        def add(self, x):
            x = int(x)
            i = np.searchsorted(self._v, x)
            if not (i < self._v.shape[0] and self._v[i] == x):
                self._v = np.insert(self._v, i, x)

        def update(self, *others):
            for o in others:
                self._v = np.union1d(self._v, _as_sorted(o))

        def discard(self, x):
            x = int(x)
            i = np.searchsorted(self._v, x)
            if i < self._v.shape[0] and self._v[i] == x:
                self._v = np.delete(self._v, i)

        def remove(self, x):
            if x not in self:
                raise KeyError(x)
            self.discard(x)

        def clear(self):
            self._v = np.zeros(0, dtype=np.int64)

        __hash__ = None  # mutable
