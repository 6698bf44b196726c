"""Time the batched scan (16 queries per pass) at BASELINE cfg 4 size.  usage: multi_query_probe.py [n_rows] [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippomm_amd.vector_ops import FeatureStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(n, 1024, dtype=torch.float32, device="cuda")
for s in range(0, n, 125_000):
    e = min(n, s + 125_000)
    blk = torch.randn(e - s, 1024, generator=g, device="cuda")
    rows[s:e] = blk / blk.norm(dim=1, keepdim=True)
store = FeatureStore(rows)
for nq in (16, 32, 1):
    q = torch.randn(nq, 1024, device="cuda")
    for _ in range(3):
        store.search_multi_device(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        store.search_multi_device(q, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    passes = (nq + 15) // 16
    print(f"{nq:3d} queries k={k}: {ms:.4f} ms  ({ms / passes:.4f} ms per pass = {n * 4096 / (ms / passes) / 1e6:.0f} GB/s, "
          f"{n * 4096 / (ms / passes) / 1e6 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)
