"""Does replaying the towers' launch sequence as a hipGraph shorten it?  (experiment, round 6)
   python tools/perf_graph_capture.py [B=200]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from seesaw_amd.models.clip import ClipModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
m = ClipModel.random_init(seed=1234, device=0)
x = torch.randn(B, 3, 224, 224, device=dev)
o = torch.empty(B, 512, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s0 = torch.cuda.current_stream(dev).cuda_stream
eager = timed(lambda: m.embed_image_dev(x.data_ptr(), B, o.data_ptr(), True, s0))
ref = o.clone()
side = torch.cuda.Stream(dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    m.embed_image_dev(x.data_ptr(), B, o.data_ptr(), True, side.cuda_stream)  # warm (allocations) on this stream
    torch.cuda.synchronize()
    try:
        g.capture_begin()
        m.embed_image_dev(x.data_ptr(), B, o.data_ptr(), True, side.cuda_stream)
        g.capture_end()
        ok = True
    except Exception as e:
        print("capture failed:", type(e).__name__, e)
        ok = False
if ok:
    o.zero_()
    replay = timed(g.replay)
    print(f"image tower B={B}: eager {eager:.3f} ms, graph replay {replay:.3f} ms, same output {bool(torch.equal(o, ref))}")
else:
    print(f"image tower B={B}: eager {eager:.3f} ms")
