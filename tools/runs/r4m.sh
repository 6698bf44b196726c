#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r4m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_scan_prefilter.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -12 $OUT/tests.log
timeout 300 python - > $OUT/prefilter_time.log 2>&1 <<'PY'
import torch, json
from hippomm_amd.vector_ops import FeatureStore
n=1_000_000
g=torch.Generator(device="cuda").manual_seed(42)
rows=torch.empty(n,1024,device="cuda")
for s in range(0,n,125000):
    b=torch.randn(125000,1024,generator=g,device="cuda"); rows[s:s+125000]=b/b.norm(dim=1,keepdim=True)
q=torch.randn(1024,generator=torch.Generator(device="cuda").manual_seed(43),device="cuda")
fs=FeatureStore(rows).build_shadow()
def t(fn,it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
st=torch.zeros(2,dtype=torch.int32,device="cuda")
for kk in (1,5,32,64):
    a=t(lambda: fs.search_device(q,kk)); b=t(lambda: fs.search_prefiltered_device(q,kk,st))
    print(json.dumps({"k":kk,"ms_exact":round(a,4),"ms_prefilter":round(b,4),"shadow_GBps":round(2.048e9/b/1e6,1),"stats":st.cpu().tolist()}))
PY
cat $OUT/prefilter_time.log
