"""Phases of prefilter_final_kernel (the one-workgroup finish of the bf16-prefilter query) from wall-clock stamps of its thread 0
(probe build): maxima read | ranked | winners' lists gathered | ranked | hot lists -> candidates | re-scored | final cut.
    python tools/prefilter_final_stamps_probe.py [out.json]"""
import ctypes as C
import json
import statistics
import sys

import torch

from probe_common import load_probe

L, lib = load_probe()
from hippomm_amd.vector_ops import FeatureStore   # noqa: E402

N, K = 1_000_000, 32
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(N, 1024, device="cuda")
for s in range(0, N, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
store = FeatureStore(rows)
store.build_shadow()
stamps = torch.zeros(8, dtype=torch.int64, device="cuda")
lib.hmm_probe_set_prefilter_stamps.restype = None
lib.hmm_probe_set_prefilter_stamps.argtypes = [C.c_void_p]
lib.hmm_probe_set_prefilter_stamps(stamps.data_ptr())
stats = torch.zeros(2, dtype=torch.int32, device="cuda")
for _ in range(100):
    store.search_prefiltered_device(q, K, stats)
names = ["maxima read", "maxima ranked", "winners' lists gathered", "k-th key ranked", "hot lists -> candidates", "re-scored", "final cut"]
acc = {n: [] for n in names}
for _ in range(40):
    for _ in range(5):
        store.search_prefiltered_device(q, K, stats)
    torch.cuda.synchronize()
    t = stamps.cpu().tolist()
    for i, n in enumerate(names):
        acc[n].append((t[i + 1] - t[i]) / 100.0)                 # 100 MHz -> us
out = {"candidates": int(stats[0]), "us_per_phase_median": {n: round(statistics.median(v), 2) for n, v in acc.items()},
       "us_total_stamped": round(sum(statistics.median(v) for v in acc.values()), 2)}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
