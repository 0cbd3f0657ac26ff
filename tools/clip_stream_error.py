"""Error of the image / text towers against transformers.CLIPModel (f32, CPU), seeded random-init weights (GPU box).

  python tools/clip_stream_error.py
Prints min cosine and max / rms |delta| of the unit vectors for both residual-stream precisions of the tile path
(ssw_clip_set_option, per handle: images bf16, text f32 -- the default; then images f32, text bf16).
"""
import os
import sys

import numpy as np
import torch
import transformers

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seesaw_amd.models.clip import ClipModel  # noqa: E402


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


torch.manual_seed(1234)
hf = transformers.CLIPModel(transformers.CLIPConfig()).eval()
ours = ClipModel.from_hf(hf)
torch.manual_seed(0)
x = torch.randn(26, 3, 224, 224)
rng = np.random.default_rng(3)
ids = rng.integers(0, 49405, size=(12, 77)).astype(np.int64)
ids[:, 0] = 49406
ids[np.arange(12), rng.integers(2, 77, size=12)] = 49407
ids[:, 76] = 49407
with torch.inference_mode():
    ref_i = hf.get_image_features(pixel_values=x)
    ref_i = unit((ref_i.pooler_output if hasattr(ref_i, "pooler_output") else ref_i).numpy())
    ref_t = hf.get_text_features(input_ids=torch.from_numpy(ids))
    ref_t = unit((ref_t.pooler_output if hasattr(ref_t, "pooler_output") else ref_t).numpy())
for flags, names in ((0, ("f32", "f32")), (3, ("bf16", "bf16"))):
    ours.set_rows(image_bf16=bool(flags & 1), text_bf16=bool(flags & 2))
    for what, ref, got, name in (("image 26", ref_i, unit(ours.embed_image(x.numpy(), normalize=False)), names[0]),
                                 ("text 12x77", ref_t, unit(ours.embed_text(ids.astype(np.int32), normalize=False)), names[1])):
        d = np.abs(got - ref)
        print(f"{what}, {name} residual rows: cos min {(got * ref).sum(1).min():.6f}  |delta| max {d.max():.2e} "
              f"rms {np.sqrt((d ** 2).mean()):.2e}", flush=True)
ours.set_rows()
