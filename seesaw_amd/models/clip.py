"""CLIP ViT-B/32 on the GPU: weight packing + ctypes front of the ssw_clip_* entry points.

The reference loads `transformers.CLIPModel.from_pretrained(<local dir>)` and calls
`get_text_features` / `get_image_features` (seesaw/models/embeddings.py:427-455,
seesaw/models/model.py:50-63).  Here the same state_dict is flattened into one blob and the
forward pass runs in libseesaw_hip.so (bf16 MFMA GEMMs, f32 LayerNorm / softmax / residuals).
"""
from __future__ import annotations

import ctypes
import struct

import numpy as np

from .. import _lib

_LAYER_KEYS = ["layer_norm1.weight", "layer_norm1.bias", "self_attn.q_proj.weight", "self_attn.q_proj.bias",
               "self_attn.k_proj.weight", "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias",
               "self_attn.out_proj.weight", "self_attn.out_proj.bias", "layer_norm2.weight", "layer_norm2.bias",
               "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"]


def pack_clip_weights(state_dict, config) -> np.ndarray:
    """transformers.CLIPModel state_dict + CLIPConfig -> the flat blob ssw_clip_create reads."""
    v, t = config.vision_config, config.text_config
    assert v.hidden_act == "quick_gelu" and t.hidden_act == "quick_gelu", "only quick_gelu CLIP variants"
    header = struct.pack("<8s6i7iifi", b"SSWCLIP1", v.hidden_size, v.num_hidden_layers, v.num_attention_heads,
                         v.intermediate_size, v.image_size, v.patch_size, t.hidden_size, t.num_hidden_layers,
                         t.num_attention_heads, t.intermediate_size, t.max_position_embeddings, t.vocab_size,
                         t.eos_token_id, config.projection_dim, float(v.layer_norm_eps), 0)
    assert len(header) == 72

    def get(name):
        x = state_dict[name]
        x = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        return np.ascontiguousarray(x, dtype=np.float32).reshape(-1)

    parts = [get("vision_model.embeddings.class_embedding"), get("vision_model.embeddings.patch_embedding.weight"),
             get("vision_model.embeddings.position_embedding.weight"), get("vision_model.pre_layrnorm.weight"),
             get("vision_model.pre_layrnorm.bias")]
    for i in range(v.num_hidden_layers):
        parts += [get(f"vision_model.encoder.layers.{i}.{k}") for k in _LAYER_KEYS]
    parts += [get("vision_model.post_layernorm.weight"), get("vision_model.post_layernorm.bias"),
              get("visual_projection.weight"), get("text_model.embeddings.token_embedding.weight"),
              get("text_model.embeddings.position_embedding.weight")]
    for i in range(t.num_hidden_layers):
        parts += [get(f"text_model.encoder.layers.{i}.{k}") for k in _LAYER_KEYS]
    parts += [get("text_model.final_layer_norm.weight"), get("text_model.final_layer_norm.bias"),
              get("text_projection.weight")]
    body = np.concatenate(parts)
    blob = np.empty(72 + body.nbytes, dtype=np.uint8)
    blob[:72] = np.frombuffer(header, dtype=np.uint8)
    blob[72:] = body.view(np.uint8)
    return blob


class ClipModel:
    def __init__(self, blob: np.ndarray, device: int = 0):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._h = ctypes.c_void_p()
        hdr = struct.unpack("<8s6i7iifi", blob[:72].tobytes())
        self.image_size, self.patch_size = hdr[5], hdr[6]
        self.max_positions, self.vocab_size, self.eos_token_id, self.projection_dim = hdr[11], hdr[12], hdr[13], hdr[14]
        _lib.call("ssw_clip_create", int(device), ctypes.c_void_p(blob.ctypes.data), ctypes.c_size_t(blob.nbytes),
                  ctypes.byref(self._h))

    @classmethod
    def from_hf(cls, hf_model, device: int = 0) -> "ClipModel":
        return cls(pack_clip_weights(hf_model.state_dict(), hf_model.config), device=device)

    @classmethod
    def random_init(cls, seed: int = 1234, device: int = 0) -> "ClipModel":
        """BASELINE.json: 'random-init CLIP weights' -- transformers.CLIPModel(CLIPConfig()) under a seed."""
        import torch
        import transformers
        torch.manual_seed(seed)
        return cls.from_hf(transformers.CLIPModel(transformers.CLIPConfig()).eval(), device=device)

    def close(self):
        if self._h:
            _lib.load().ssw_clip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def embed_image(self, pixel_values: np.ndarray, normalize: bool = True) -> np.ndarray:
        x = np.ascontiguousarray(pixel_values, dtype=np.float32)
        assert x.ndim == 4 and x.shape[1:] == (3, self.image_size, self.image_size), x.shape
        out = np.empty((x.shape[0], self.projection_dim), dtype=np.float32)
        _lib.call("ssw_clip_embed_image", self._h, ctypes.c_void_p(x.ctypes.data), x.shape[0], int(normalize),
                  ctypes.c_void_p(out.ctypes.data))
        return out

    def embed_tiles_u8(self, tiles: np.ndarray, normalize: bool = True) -> np.ndarray:
        """uint8 HWC tiles [B, S, S, 3] straight from the tiler; batch_tx's normalisation runs on the GPU."""
        x = np.ascontiguousarray(tiles, dtype=np.uint8)
        assert x.ndim == 4 and x.shape[1:] == (self.image_size, self.image_size, 3), x.shape
        out = np.empty((x.shape[0], self.projection_dim), dtype=np.float32)
        _lib.call("ssw_clip_embed_tiles_u8", self._h, ctypes.c_void_p(x.ctypes.data), x.shape[0], int(normalize),
                  ctypes.c_void_p(out.ctypes.data))
        return out

    def embed_image_dev(self, pixels_ptr: int, b: int, out_ptr: int, normalize: bool = True, stream_ptr: int = 0):
        _lib.call("ssw_clip_embed_image_dev", self._h, ctypes.c_void_p(stream_ptr) if stream_ptr else None,
                  ctypes.c_void_p(pixels_ptr), int(b), int(normalize), ctypes.c_void_p(out_ptr))

    def sync(self):
        _lib.call("ssw_clip_sync", self._h)

    # ssw_clip_set_option (include/seesaw_hip.h): per-handle form of the towers' tile path
    OPT_IMAGE_ROWS_BF16, OPT_TEXT_ROWS_BF16, OPT_ATTN_DIRECT, OPT_ATTN_OUT_UNFUSED, OPT_FULL_LAST_LAYER = 0, 1, 2, 3, 4

    def set_option(self, option: int, value: bool):
        _lib.call("ssw_clip_set_option", self._h, int(option), int(bool(value)))

    def set_rows(self, image_bf16: bool = False, text_bf16: bool = False):
        """residual-row precision of the two towers' tile paths (default: f32 rows in both)"""
        self.set_option(self.OPT_IMAGE_ROWS_BF16, image_bf16)
        self.set_option(self.OPT_TEXT_ROWS_BF16, text_bf16)

    def embed_text(self, input_ids: np.ndarray, normalize: bool = False) -> np.ndarray:
        ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        assert ids.ndim == 2
        out = np.empty((ids.shape[0], self.projection_dim), dtype=np.float32)
        _lib.call("ssw_clip_embed_text", self._h, ctypes.c_void_p(ids.ctypes.data), ids.shape[0], ids.shape[1],
                  int(normalize), ctypes.c_void_p(out.ctypes.data))
        return out
