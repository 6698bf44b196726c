"""Grid size of the exact scan's streaming kernels (probe build knob g_scan_blocks): the pure reader of round 2 streamed 0.889 of 8 TB/s
from 1024 workgroups and 0.847 from 2048 (profiles/r2_hbm_read.json) -- does the scan follow?  Interleaved, one process; whole queries
checked against the 2048-workgroup result.
    python tools/scan_blocks_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, setter

L, lib = load_probe()
from hippomm_amd.vector_ops import EventStore, FeatureStore   # noqa: E402

N, K = 1_000_000, 32
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(N, 1024, device="cuda")
for s in range(0, N, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
store = FeatureStore(rows)
events = EventStore.from_device_rows(rows, [500] * 2000)
cand = torch.empty(2048 * K, dtype=torch.int64, device="cuda")


def timed(fn, iters=60):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


store.build_shadow()
events.build_shadow()
knobs = {"g_scan_blocks": ("exact_query", lambda: store.search_device(q, K)),
         "g_sims_blocks": ("per_event", lambda: events.search_segments_device(q, events.offsets, 5)),
         "g_prefilter_blocks": ("prefilter_query", lambda: store.search_prefiltered_device(q, K)),
         "g_prefilter_sims_blocks": ("per_event_prefilter", lambda: events.search_segments_device(q, events.offsets, 5, prefilter=True))}
# g_sims_blocks / g_prefilter_sims_blocks size the similarity passes of UNTIL round 6 (scan_sims_kernel, prefilter_sims_kernel), which
# only the probe build still has: switch to them for this sweep (the shipped deferred-store passes run on g_scan_blocks / g_prefilter_blocks)
setter(lib, "g_sims_deferred")(0)
setter(lib, "g_prefilter_sims_deferred")(0)
out = {}
for knob, (leg, fn) in knobs.items():
    set_k = setter(lib, knob)
    set_k(2048)
    ref = [t.clone() for t in fn()]
    grid = (2048, 1024, 768, 640, 512, 448, 384, 320, 256)
    for rep in range(2):
        for blocks in grid:
            set_k(blocks)
            rec = out.setdefault(f"{leg}@{blocks}", {"ms": []})
            rec["ms"].append(round(timed(fn), 4))
            got = fn()
            rec["same_result"] = all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
                                     for a, b in zip(got, ref))
    set_k(2048)
    print(leg, {b: min(out[f"{leg}@{b}"]["ms"]) for b in grid}, all(out[f"{leg}@{b}"]["same_result"] for b in grid), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
