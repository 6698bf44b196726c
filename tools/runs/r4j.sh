#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r4j
mkdir -p $OUT
timeout 300 python - > $OUT/multi_shadow.log 2>&1 <<'PY'
import torch, json
from hippomm_amd.vector_ops import FeatureStore
n=1_000_000
g=torch.Generator(device="cuda").manual_seed(42)
rows=torch.empty(n,1024,device="cuda")
for s in range(0,n,125000):
    b=torch.randn(125000,1024,generator=g,device="cuda"); rows[s:s+125000]=b/b.norm(dim=1,keepdim=True)
q=torch.randn(1024,generator=torch.Generator(device="cuda").manual_seed(43),device="cuda")
q16=torch.randn(16,1024,generator=torch.Generator(device="cuda").manual_seed(44),device="cuda")
fs=FeatureStore(rows)
def t(fn,it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
print("multi before shadow", t(lambda: fs.search_multi_device(q16,32)), "single", t(lambda: fs.search_device(q,32)))
fs.build_shadow(); torch.cuda.synchronize()
print("multi after shadow", t(lambda: fs.search_multi_device(q16,32)), "single", t(lambda: fs.search_device(q,32)), "prefilter", t(lambda: fs.search_prefiltered_device(q,32)))
print("multi again", t(lambda: fs.search_multi_device(q16,32)))
fs._shadow=None; torch.cuda.empty_cache()
print("multi after freeing shadow", t(lambda: fs.search_multi_device(q16,32)))
PY
cat $OUT/multi_shadow.log
