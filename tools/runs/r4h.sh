#!/bin/bash
# GPU session r4h: per-event prefilter parity + timing; joules-per-frame account
REPO=$PWD
OUT=$REPO/gpurun_out/r4h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_scan_prefilter.py tests/test_gpu_segments.py tests/test_gpu_retrieval.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -15 $OUT/tests.log
timeout 300 python - > $OUT/seg_time.log 2>&1 <<'PY'
import torch, json
from hippomm_amd.vector_ops import EventStore
n=1_000_000
g=torch.Generator(device="cuda").manual_seed(42)
rows=torch.empty(n,1024,device="cuda")
for s in range(0,n,125000):
    b=torch.randn(125000,1024,generator=g,device="cuda"); rows[s:s+125000]=b/b.norm(dim=1,keepdim=True)
q=torch.randn(1024,generator=torch.Generator(device="cuda").manual_seed(43),device="cuda")
def t(fn,it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
for n_ev in (2000, 250, 20000):
    es=EventStore.from_device_rows(rows,[n//n_ev]*n_ev); es.build_shadow()
    for k in (5,32):
        a=t(lambda: es.search_segments_device(q,es.offsets,k)); b=t(lambda: es.search_segments_device(q,es.offsets,k,prefilter=True))
        print(json.dumps({"events":n_ev,"k":k,"ms_exact":round(a,4),"ms_prefilter":round(b,4)}))
PY
cat $OUT/seg_time.log
timeout 300 python tools/energy_account_probe.py $OUT/energy_account.json > $OUT/energy_account.log 2>&1
tail -60 $OUT/energy_account.log
