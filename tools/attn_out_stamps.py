"""in-kernel phase stamps of the fused attention + out-projection launch (lab build, SSW_AO_STAMPS=1):
   SSW_AO_STAMPS=1 [SSW_CLIP_BF16_STREAM=1] python tools/attn_out_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seesaw_amd.models.clip import ClipModel
from seesaw_amd import _lib
_lib.debug_hooks().__enter__()
m = ClipModel.random_init(seed=1234)
B = int(os.environ.get('AO_B', '200'))
x = torch.randn(B, 3, 224, 224, device="cuda")
out = torch.empty(B, 512, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    m.embed_image_dev(x.data_ptr(), B, out.data_ptr(), True, s)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(1024 * 32, dtype=np.uint64)
lib.ssw_debug_attn_out_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1024 * 32)
allst = buf.reshape(1024, 32).astype(np.int64)
allst = allst[allst[:, 22] > 0]   # workgroups that took an image (the affinity launch has a few idle ones)
assert allst.shape[0] == B and sorted((allst[:, 22] - 1).tolist()) == list(range(B)), allst.shape
tile_of = ((allst[:, 22] - 1) * 50 + 25) // 128
print("XCC of the workgroups (first 16 launched):", allst[:16, 21].tolist(), "images:", (allst[:16, 22] - 1).tolist())
per = (((B * 50 + 127) // 128) + 7) // 8
print("workgroups per XCC:", np.bincount(allst[:, 21], minlength=8).tolist(), " images whose row-tile run (contiguous map) is on the workgroup's XCC id if block 0 = XCC of run 0:",
      int((tile_of // per == (allst[:, 21] - allst[0, 21]) % 8).sum()))
it = allst[:, 8:20] - allst[:, :1]
print("head pairs, cycles since start [staged, computed] x 6 (median):", np.median(it, axis=0).astype(int).tolist())
st = allst[:, :5]
d = np.diff(st, axis=1)
print("median cycles per phase [attention, product, stores, stats]:", np.median(d, axis=0), "total", np.median(st[:, 4] - st[:, 0]))
print("p10/p90 total", np.percentile(st[:, 4] - st[:, 0], [10, 90]))
print("start spread (cycles)", st[:, 0].max() - st[:, 0].min(), "end spread", st[:, 4].max() - st[:, 4].min(), "launch span", st[:, 4].max() - st[:, 0].min())
