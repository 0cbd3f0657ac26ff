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
        """two interchangeable representations, each built from the other on demand: a sorted int64 array
        (`_v`: ordered iteration, rank, numpy set algebra) and a python set (`_s`: O(1) membership / add -- the
        session's bookkeeping adds one id at a time, every round)"""
        __slots__ = ("_arr", "_set")

        def __init__(self, values: Iterable[int] = None):
            self._arr = _as_sorted(values)
            self._set = None

        @property
        def _v(self) -> np.ndarray:
            if self._arr is None:
                self._arr = np.fromiter(self._set, dtype=np.int64, count=len(self._set))
                self._arr.sort()
            return self._arr

        @property
        def _s(self) -> set:
            if self._set is None:
                self._set = set(self._arr.tolist())
            return self._set

        # ---- queries
        def __len__(self):
            return len(self._set) if self._arr is None else int(self._arr.shape[0])

        def __iter__(self):
            return iter(self._v.tolist())

        def __contains__(self, x):
            x = int(x)
            if self._set is not None:
                return x in self._set
            i = np.searchsorted(self._arr, x)
            return bool(i < self._arr.shape[0] and self._arr[i] == x)

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
            out._arr = arr
            out._set = None
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
            out = self._new(None if self._arr is None else self._arr.copy())
            out._set = None if self._set is None else set(self._set)
            return out

    class FrozenBitMap(_IntSet):
        """immutable"""

    class BitMap(_IntSet):
        def add(self, x):
            s = self._s
            x = int(x)
            if x not in s:
                s.add(x)
                self._arr = None  # rebuilt (sorted) when an ordered view is next asked for

        def update(self, *others):
            for o in others:
                self._arr = np.union1d(self._v, _as_sorted(o))
                self._set = None

        def discard(self, x):
            s = self._s
            x = int(x)
            if x in s:
                s.discard(x)
                self._arr = None

        def remove(self, x):
            if x not in self:
                raise KeyError(x)
            self.discard(x)

        def clear(self):
            self._arr = np.zeros(0, dtype=np.int64)
            self._set = None

        __hash__ = None  # mutable
