"""Is bench.py's slow batched_16_queries (0.77 ms vs 0.66 in a fresh process) the clock state the board is left in after seconds of
power-capped tower forwards?  5 s of forwards, then the multi-query pass timed every ~10 ms for a while, a 3 s pause, again; the
single-query scan the same way.  usage: multi_thermal_probe.py [out.json]"""
import json
import sys
import time

import torch

from probe_common import ROOT, own_power_file  # noqa: F401
import bench
from hippomm_amd.encoder import HipTower, synthetic_state_dict
from hippomm_amd.vector_ops import FeatureStore

N, K = 1_000_000, 32
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(N, 1024, device="cuda")
for s in range(0, N, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
q = torch.randn(1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(43))
q16 = torch.randn(16, 1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(44))
store = FeatureStore(rows)
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
frames = bench.synthetic_frames(0, 256, "cuda")
emb = torch.empty(256, 1024, device="cuda")
pf = own_power_file()
out = []


def watts():
    try:
        return round(int(open(pf).read()) / 1e6, 1) if pf else None
    except Exception:
        return None


def series(tag, fn, n=12):
    vals = []
    for _ in range(n):
        vals.append(round(bench.event_time_ms(fn, 10, warmup=0), 4))
    rec = {"leg": tag, "ms_per_10_iterations": vals, "watts_after": watts()}
    out.append(rec)
    print(json.dumps(rec), flush=True)


def heat(seconds):
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(5):
            tower.forward_into(frames, emb)
        torch.cuda.synchronize()


series("cold: multi16", lambda: store.search_multi_device(q16, K))
series("cold: single", lambda: store.search_device(q, K))
heat(6.0)
series("right after 6 s of batch-256 forwards: multi16", lambda: store.search_multi_device(q16, K))
heat(6.0)
series("right after 6 s of forwards: single", lambda: store.search_device(q, K))
series("... then multi16", lambda: store.search_multi_device(q16, K))
time.sleep(3.0)
series("after a 3 s pause: multi16", lambda: store.search_multi_device(q16, K))
heat(6.0)
del tower
torch.cuda.empty_cache()
series("6 s of forwards, tower freed: multi16", lambda: store.search_multi_device(q16, K))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
