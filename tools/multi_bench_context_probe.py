"""batched_16_queries is 0.65-0.67 ms in a fresh process (multi_shadow_probe.py) and 0.76-0.79 ms inside bench.py (five round-4
sessions, one round-5 session).  What in bench.py's history does it: the tower legs before the scan (allocator / page placement of
the 4-GB store) or the legs inside scan_bench?  usage: multi_bench_context_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import ROOT  # noqa: F401
import bench
from hippomm_amd.encoder import HipTower, synthetic_state_dict
from hippomm_amd.vector_ops import FeatureStore

N, K = 1_000_000, 32
out = []


def t_ms(fn, iters=30, warmup=5):
    return round(bench.event_time_ms(fn, iters, warmup=warmup), 4)


def emit(**kw):
    out.append(kw)
    print(json.dumps(kw), flush=True)


def make_rows():
    g = torch.Generator(device="cuda").manual_seed(42)
    rows = torch.empty(N, 1024, device="cuda")
    for s in range(0, N, 125_000):
        blk = torch.randn(125_000, 1024, generator=g, device="cuda")
        rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
    return rows


q = torch.randn(1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(43))
q16 = torch.randn(16, 1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(44))

# 1. fresh process
rows = make_rows()
store = FeatureStore(rows)
emit(leg="fresh process: multi16", ms=t_ms(lambda: store.search_multi_device(q16, K)), ptr=hex(rows.data_ptr()))
emit(leg="fresh process: single", ms=t_ms(lambda: store.search_device(q, K)))
del store, rows
torch.cuda.empty_cache()

# 2. what bench.py does before the scan: tower, frames, 20 + 5 forwards, the kernel table, then free everything
sd = synthetic_state_dict(("vision",), seed=1234)
tower = HipTower("vision", sd)
del sd
torch.cuda.empty_cache()
frames = bench.synthetic_frames(0, 256, "cuda")
emb = torch.empty(256, 1024, device="cuda")
for _ in range(10):
    tower.forward_into(frames, emb)
torch.cuda.synchronize()
bench.gemm_roofline(256)
del tower, frames, emb
torch.cuda.empty_cache()
rows = make_rows()
store = FeatureStore(rows)
emit(leg="after the tower legs: multi16", ms=t_ms(lambda: store.search_multi_device(q16, K)), ptr=hex(rows.data_ptr()))
emit(leg="after the tower legs: single", ms=t_ms(lambda: store.search_device(q, K)))
# 3. scan_bench's own order
lib = bench  # noqa
cand = torch.empty(2048 * K, dtype=torch.int64, device="cuda")
store.build_shadow()
emit(leg="+ shadow built: multi16", ms=t_ms(lambda: store.search_multi_device(q16, K)))
emit(leg="prefilter", ms=t_ms(lambda: store.search_prefiltered_device(q, K)))
emit(leg="after prefilter: multi16 (30 it)", ms=t_ms(lambda: store.search_multi_device(q16, K)))
emit(leg="after prefilter: multi16 (10 it, warmup 3)", ms=t_ms(lambda: store.search_multi_device(q16, K), 10, 3))
# 4. the same rows in a NEW allocation
rows2 = rows.clone()
store2 = FeatureStore(rows2)
emit(leg="cloned store (new allocation): multi16", ms=t_ms(lambda: store2.search_multi_device(q16, K)), ptr=hex(rows2.data_ptr()))
del store2, rows2
# 5. workspace of its own (search_multi_device shares self._ws with the single-query scan)
store._ws = None
emit(leg="fresh workspace: multi16", ms=t_ms(lambda: store.search_multi_device(q16, K)))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
