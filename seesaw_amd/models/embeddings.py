"""Embedding front ends with the reference's XEmbedding interface
(seesaw/models/embeddings.py:427-466): `from_string(string=...) -> [1, 512]`,
`from_image(preprocessed_image=...) -> [B, 512]`, backed by the HIP CLIP towers."""
from __future__ import annotations

import os
import zlib

import numpy as np

from .clip import ClipModel


class HashTokenizer:
    """Deterministic stand-in used when no CLIP vocabulary files are available (as in the
    build image): words hash into the id range; sequence = [BOS, ids..., EOS]."""

    def __init__(self, vocab_size: int, bos: int = 49406, eos: int = 49407, max_len: int = 77):
        self.vocab_size, self.bos, self.eos, self.max_len = vocab_size, bos, eos, max_len

    def __call__(self, text: str) -> np.ndarray:
        words = text.lower().split()[: self.max_len - 2]
        ids = [zlib.crc32(w.encode()) % (self.bos - 1) for w in words]
        return np.array([[self.bos] + ids + [self.eos]], dtype=np.int32)


def _device_index(device) -> int:
    if isinstance(device, str):
        return int(device.split(":")[-1]) if ":" in device else 0
    return int(device)


class HGWrapper:
    """text side (HGWrapper.from_string): un-normalised text features, cached per string."""

    def __init__(self, path=None, device=0, num_cpus=1, model: ClipModel = None, tokenizer=None):
        self.model = model if model is not None else load_clip(path, device=_device_index(device))
        self.tokenizer = tokenizer or _load_tokenizer(path, self.model)
        self.string_cache = {}

    def ready(self):
        return True

    def from_string(self, *, string=None, str_vec=None, numpy=True):
        if str_vec is not None:
            return str_vec
        if string not in self.string_cache:
            ids = self.tokenizer(string)
            self.string_cache[string] = self.model.embed_text(ids, normalize=False).reshape(1, -1)
        return self.string_cache[string]

    def from_image(self, *, preprocessed_image=None, image=None, img_vec=None, numpy=True, pooled=True):
        if img_vec is not None:
            return img_vec
        x = preprocessed_image
        x = x.detach().cpu().float().numpy() if hasattr(x, "detach") else np.asarray(x, dtype=np.float32)
        return self.model.embed_image(x, normalize=True)


def gen_strided_blocks(vecs, width_size, stride_size, flatten=True):
    """every width_size x width_size window of the last two axes at the given stride, row-major over (ii, jj), stacked
    along the batch axis (seesaw/models/embeddings.py:252-281; its `center` switch is hard-wired off there)"""
    assert flatten
    h, w = vecs.shape[-2:]
    iis = list(range(0, h - width_size + 1, stride_size))
    jjs = list(range(0, w - width_size + 1, stride_size))
    cuts = [vecs[..., ii:ii + width_size, jj:jj + width_size] for ii in iis for jj in jjs]
    return np.concatenate(cuts), iis, jjs


class SlidingWindow:
    """the kernel applied to every strided window of ONE image; output [1, *kernel output dims, len(iis), len(jjs)]
    (seesaw/models/embeddings.py:344-378)"""

    def __init__(self, kernel, kernel_size, stride=None, center=False):
        self.kernel, self.kernel_size = kernel, kernel_size
        self.stride = stride if stride is not None else kernel_size
        self.center = center

    def __call__(self, tensor):
        return self.forward(tensor)

    def forward(self, tensor):
        assert tensor.shape[0] == 1, "just do this"
        assert len(tensor.shape) == 4, "also"
        input_batch, iis, jjs = gen_strided_blocks(tensor, self.kernel_size, self.stride, flatten=True)
        output_batch = np.asarray(self.kernel(np.ascontiguousarray(input_batch)))
        val_shape = output_batch.shape[1:]
        v = np.moveaxis(output_batch, 0, -1)  # ij last
        return v.reshape((1,) + val_shape + (len(iis), len(jjs)))


class ImageEmbedding:
    """image side (seesaw/models/model.py:67-89): L2-normalised features of a batch of 224 x 224 tiles, or -- with
    add_slide, the reference's default outside the indexing job -- of every half-overlapping 224 x 224 window of one
    larger image (SlidingWindow(kernel_size=224, stride=112))."""

    def __init__(self, device=0, jit_path=None, add_slide=False, model: ClipModel = None):
        self.model = model if model is not None else load_clip(jit_path, device=_device_index(device))
        self.add_slide = bool(add_slide)
        kernel_size = 224  # (changes with the variant, model.py:77)
        self._slide = SlidingWindow(lambda x: self.model.embed_image(x, normalize=True), kernel_size=kernel_size,
                                    stride=kernel_size // 2, center=True) if add_slide else None

    def __call__(self, *, preprocessed_image):
        return self.forward(preprocessed_image=preprocessed_image)

    def forward(self, *, preprocessed_image):
        x = preprocessed_image
        x = x.detach().cpu().float().numpy() if hasattr(x, "detach") else np.asarray(x, dtype=np.float32)
        if self._slide is not None:
            return self._slide(x)
        return self.model.embed_image(x, normalize=True)


SYNTHETIC_PREFIX = "synthetic:"  # explicit opt-in: "synthetic:random-init" or "synthetic:random-init:<seed>"


def _synthetic_seed(path):
    """seed when `path` is the explicit synthetic opt-in, else None"""
    if isinstance(path, str) and path.startswith(SYNTHETIC_PREFIX):
        parts = path.split(":")
        return int(parts[2]) if len(parts) > 2 else 1234
    return None


def _load_tokenizer(path, model: ClipModel):
    """CLIPTokenizer of the model directory (embeddings.py:433-434 loads both from one path).  The hash
    stand-in is used only with the explicit synthetic opt-in (or no path at all, bench / tests): a real
    model directory without its vocabulary raises, as from_pretrained does in the reference."""
    if path is None or _synthetic_seed(path) is not None:
        return HashTokenizer(model.vocab_size, eos=model.eos_token_id if model.eos_token_id != 2 else model.vocab_size - 1,
                             max_len=model.max_positions)
    if not os.path.exists(os.path.join(str(path), "vocab.json")):
        raise FileNotFoundError(f"no CLIP tokenizer files (vocab.json) under {path!r}")
    import transformers
    tok = transformers.CLIPTokenizer.from_pretrained(path)
    return lambda s: np.asarray(tok(s, return_tensors="np")["input_ids"], dtype=np.int32)


_CLIP_CACHE = {}


def load_clip(path=None, device: int = 0) -> ClipModel:
    """HF CLIP directory (config + weights) -> ClipModel.  Seeded random-init weights (BASELINE.json's
    synthetic configuration) only on request: path None or "synthetic:random-init[:seed]".  Anything else
    that cannot be loaded raises FileNotFoundError -- an index whose info.json names a missing model must
    not serve meaningless text vectors."""
    key = (path, device)
    if key not in _CLIP_CACHE:
        seed = 1234 if path is None else _synthetic_seed(path)
        if seed is not None:
            _CLIP_CACHE[key] = ClipModel.random_init(seed=seed, device=device)
        elif os.path.isdir(str(path)) and os.path.exists(os.path.join(str(path), "config.json")):
            import transformers
            hf = transformers.CLIPModel.from_pretrained(path).eval()
            _CLIP_CACHE[key] = ClipModel.from_hf(hf, device=device)
        else:
            raise FileNotFoundError(f"CLIP model directory {path!r} not found (expected config.json + weights); "
                                    f"pass '{SYNTHETIC_PREFIX}random-init' for seeded random weights")
    return _CLIP_CACHE[key]


def load_embedding(model_path=None, device: int = 0):
    if model_path is None:
        return None
    return HGWrapper(model_path, device=device)
