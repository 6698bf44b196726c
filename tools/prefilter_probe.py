"""The fp32 scan against the bf16-prefilter scan at BASELINE cfg 4's size (1M x 1024 rows), single query at k = 1 / 5 / 32 / 64 and
per event at 2000 / 250 / 20 000 events: HIP-event time per query, candidates re-scored, identity of the results.
usage: prefilter_probe.py [json_out]   (product library: no probe build needed)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippomm_amd.vector_ops import EventStore, FeatureStore

n = 1_000_000
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(n, 1024, device="cuda")
for s in range(0, n, 125000):
    b = torch.randn(125000, 1024, generator=g, device="cuda")
    rows[s:s + 125000] = b / b.norm(dim=1, keepdim=True)
q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")


def t(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


res = []
fs = FeatureStore(rows).build_shadow()
st = torch.zeros(2, dtype=torch.int32, device="cuda")
for k in (1, 5, 32, 64):
    a, b = t(lambda: fs.search_device(q, k)), t(lambda: fs.search_prefiltered_device(q, k, st))
    i0, s0 = fs.search_device(q, k)
    i1, s1 = fs.search_prefiltered_device(q, k, st)
    rec = {"scan": "single query", "k": k, "ms_fp32": round(a, 4), "ms_prefilter": round(b, 4), "shadow_GBps": round(2.048e9 / b / 1e6, 1),
           "candidates_rescored": int(st[0]), "saturated_lists": int(st[1]),
           "identical": bool(torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)))}
    res.append(rec)
    print(json.dumps(rec), flush=True)
for n_ev in (2000, 250, 20000):
    es = EventStore.from_device_rows(rows, [n // n_ev] * n_ev)
    es._shadow = fs._shadow
    for k in (5, 32):
        a = t(lambda: es.search_segments_device(q, es.offsets, k), 20)
        b = t(lambda: es.search_segments_device(q, es.offsets, k, prefilter=True), 20)
        x, y = es.search_segments_device(q, es.offsets, k), es.search_segments_device(q, es.offsets, k, prefilter=True)
        rec = {"scan": "per event", "events": n_ev, "k": k, "ms_fp32": round(a, 4), "ms_prefilter": round(b, 4),
               "identical": bool(torch.equal(x[0], y[0]) and torch.equal(x[1].view(torch.int32), y[1].view(torch.int32)))}
        res.append(rec)
        print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
