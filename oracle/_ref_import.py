"""Import the reference (orm011/seesaw, read-only at /root/reference) inside the BUILD
container so that oracle/gen_golden.py can run its functions on seeded inputs.

Tool for generating tests/golden/*.npz only: nothing here travels as reference code, the
reference is never copied, and neither tests nor the product import this module on the
GPU box (where /root/reference does not exist).

Several of the reference's module-level imports are packages this image does not ship
(ray, pyroaring, annoy, pynndescent, torchvision, pytorch_lightning...).  None of them is
on the numeric path being pinned; they are cluster plumbing (ray), an approximate index we
replace (annoy / pynndescent) or image preprocessing (torchvision).  To let the pure-numpy
/ scipy / torch code of the reference import, empty stand-in modules are registered for
them; `pyroaring.BitMap` is the one stand-in with behaviour (a sorted integer set: ordered
iteration, rank, set algebra), because `_get_top_dbidxs` / `CoarseIndex.query` call it.
pydantic is aliased to its bundled v1 API, which is what the reference was written for.
"""
import importlib
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"


class BitMap:
    """Minimal sorted-integer-set stand-in for pyroaring.BitMap / FrozenBitMap."""

    def __init__(self, values=None):
        self._s = set(int(v) for v in values) if values is not None else set()

    # set algebra
    def difference(self, other):
        return BitMap(self._s - set(other))

    def union(self, *others):
        s = set(self._s)
        for o in others:
            s |= set(int(v) for v in o)
        return BitMap(s)

    def intersection(self, other):
        return BitMap(self._s & set(int(v) for v in other))

    def intersection_cardinality(self, other):
        return len(self._s & set(int(v) for v in other))

    def __sub__(self, other):
        return self.difference(other)

    def __or__(self, other):
        return self.union(other)

    def __and__(self, other):
        return self.intersection(other)

    def update(self, values):
        self._s.update(int(v) for v in values)

    def add(self, v):
        self._s.add(int(v))

    def clear(self):
        self._s.clear()

    def copy(self):
        return BitMap(self._s)

    def rank(self, v):
        v = int(v)
        return sum(1 for x in self._s if x <= v)

    def __contains__(self, v):
        return int(v) in self._s

    def __len__(self):
        return len(self._s)

    def __iter__(self):
        return iter(sorted(self._s))

    def __eq__(self, other):
        return set(self) == set(other)

    def __hash__(self):
        return hash(frozenset(self._s))

    def __array__(self, dtype=None, copy=None):
        import numpy as np
        return np.array(sorted(self._s), dtype=dtype or np.int64)

    def __repr__(self):
        return f"BitMap({sorted(self._s)[:8]}{'...' if len(self._s) > 8 else ''})"


class _Anything:
    """Attribute sink used for decorators / classes pulled from stand-in modules."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # used as a decorator
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    m.__dict__.update(attrs)

    def _getattr(attr, _m=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything()

    m.__getattr__ = _getattr
    sys.modules[name] = m
    return m


def install():
    if "seesaw" in sys.modules:
        return
    import transformers  # noqa: F401  (real one first; it probes torchvision lazily)
    import transformers.models.clip.modeling_clip  # noqa: F401
    import pydantic.v1 as pydantic_v1

    class TensorArray(list):  # ray.data.extensions.TensorArray: a list of row arrays here
        def __init__(self, arr):
            import numpy as np
            super().__init__(list(np.asarray(arr)))

        def to_numpy(self):
            import numpy as np
            return np.stack(self) if len(self) else np.zeros((0, 0))

    for name in ["ray", "ray.actor", "ray.util", "ray.data", "ray.data.extensions",
                 "ray.data.datasource", "ray.data.datasource.file_meta_provider",
                 "pynndescent", "annoy", "torchvision", "torchvision.transforms",
                 "torchvision.models", "torchvision.ops", "torchvision.ops.boxes",
                 "pytorch_lightning", "clip", "ftfy", "shapely", "shapely.geometry"]:
        _stub(name)
    sys.modules["ray.data.extensions"].TensorArray = TensorArray

    # torchvision.ops.boxes.{box_area,_box_inter_union}: the two third-party functions the
    # reference's box_iou calls (seesaw/box_utils.py:336-350), per torchvision's documented
    # semantics (boxes as x1,y1,x2,y2; intersection clamped at 0).
    import torch

    def box_area(b):
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def _box_inter_union(b1, b2):
        a1, a2 = box_area(b1), box_area(b2)
        lt = torch.max(b1[:, None, :2], b2[:, :2])
        rb = torch.min(b1[:, None, 2:], b2[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, :, 0] * wh[:, :, 1]
        return inter, a1[:, None] + a2 - inter

    # torchvision.transforms.Normalize, the one transform batch_tx applies (multiscale_tools.py:167-183), per its documented
    # semantics: output[c] = (input[c] - mean[c]) / std[c], mean / std as tensors of the input's dtype
    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return t.clone().sub_(mean).div_(std)

    sys.modules["torchvision.transforms"].Normalize = Normalize
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    for modname in ("torchvision.ops", "torchvision.ops.boxes"):
        sys.modules[modname].box_area = box_area
        sys.modules[modname]._box_inter_union = _box_inter_union
    sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]
    sys.modules["torchvision.ops"].boxes = sys.modules["torchvision.ops.boxes"]
    sys.modules["ray"].remote = _Anything()
    _stub("pyroaring", BitMap=BitMap, FrozenBitMap=BitMap)
    sys.modules["pydantic"] = pydantic_v1
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def ref(module: str):
    install()
    return importlib.import_module(module)
