"""one text query of 8 tokens, host ids in -> host embedding out, 50 repetitions (tools/collect_profiles.sh traces it)"""
import time, numpy as np
from seesaw_amd.models.clip import ClipModel
m=ClipModel.random_init(seed=1234)
ids=np.random.default_rng(0).integers(0,49405,(1,8)).astype(np.int32); ids[:,0]=49406; ids[:,-1]=49407
m.embed_text(ids)
t0=time.perf_counter()
for _ in range(50): m.embed_text(ids)
print("1x8 tokens: %.3f ms"%((time.perf_counter()-t0)/50*1e3))
