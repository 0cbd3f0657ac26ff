"""CLIP image tower at B=200 only (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from seesaw_amd import _lib
from seesaw_amd.models.clip import ClipModel

if len(sys.argv) > 2:  # GEMM variants are named: the lab build (ssw_tune_gemm lives there); otherwise the product library
    _lib.debug_hooks().__enter__()
m = ClipModel.random_init(seed=1234)
if os.environ.get("SSW_CLIP_ROWS_BF16"):
    m.set_rows(image_bf16=True)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = torch.randn(B, 3, 224, 224, device=dev)
out = torch.empty(B, 512, device=dev)
s = torch.cuda.current_stream().cuda_stream

variants = [int(v) for v in sys.argv[2:]] or [-1]
for v in variants:
    if v >= 0:
        _lib.call("ssw_tune_gemm", v)
    for _ in range(3):
        m.embed_image_dev(x.data_ptr(), B, out.data_ptr(), True, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.embed_image_dev(x.data_ptr(), B, out.data_ptr(), True, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B} gemm variant {v}: {dt*1e3:.3f} ms, {B*8.818/dt/1e3:.1f} TFLOP/s", flush=True)
