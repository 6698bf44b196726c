"""Round-4 verdict item 2a: `batched_16_queries` went 0.672 -> 0.765 ms between rounds 3 and 4 with cosine_topk_multi.hip untouched;
round 4's bench times it right after store.build_shadow() and the prefilter leg.  One session, 50 iterations per leg, the legs in
the order bench.py runs them and in others: is it the 2-GB shadow's placement, the order, or the 10-iteration sample?
usage: multi_shadow_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import ROOT  # noqa: F401  (sys.path)
from hippomm_amd.vector_ops import FeatureStore

N, K = 1_000_000, 32
out = {"rows": N, "k": K, "legs": []}


def times(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    # and back to back, as bench.py times it
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return {"min": round(ts[0], 4), "median": round(ts[len(ts) // 2], 4), "p90": round(ts[int(len(ts) * 0.9)], 4),
            "back_to_back_mean": round(e0.elapsed_time(e1) / iters, 4)}


def leg(name, fn, **kw):
    rec = {"leg": name, **times(fn, **kw)}
    rec["hbm_frac_median"] = round(N * 4096 / rec["median"] / 1e6 / 8000, 4)
    out["legs"].append(rec)
    print(json.dumps(rec), flush=True)


g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(N, 1024, device="cuda")
for s in range(0, N, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
q = torch.randn(1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(43))
q16 = torch.randn(16, 1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(44))
store = FeatureStore(rows)
out["store_ptr"] = hex(rows.data_ptr())

leg("multi16 first thing", lambda: store.search_multi_device(q16, K))
leg("single query", lambda: store.search_device(q, K))
leg("multi16 after single queries", lambda: store.search_multi_device(q16, K))
store.build_shadow()
torch.cuda.synchronize()
out["shadow_ptr"] = hex(store._shadow.data_ptr())
leg("multi16 right after build_shadow", lambda: store.search_multi_device(q16, K))
leg("prefilter query", lambda: store.search_prefiltered_device(q, K))
leg("multi16 after the prefilter leg (bench.py's order)", lambda: store.search_multi_device(q16, K))
leg("multi16, 10 iterations as bench.py r4", lambda: store.search_multi_device(q16, K), iters=10, warmup=3)
store._shadow = None
torch.cuda.empty_cache()
leg("multi16 after dropping the shadow", lambda: store.search_multi_device(q16, K))
pad = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")           # something else where the shadow was
leg("multi16 beside 3 GiB of untouched padding", lambda: store.search_multi_device(q16, K))
del pad
leg("multi16 k=5", lambda: store.search_multi_device(q16, 5))
leg("single query again", lambda: store.search_device(q, K))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
