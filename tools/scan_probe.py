"""Time the scan at BASELINE cfg 4 (1M x 1024 fp32, k=32) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippomm_amd.vector_ops import FeatureStore

n, k = 1_000_000, 32
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(n, 1024, dtype=torch.float32, device="cuda")
for s in range(0, n, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
q = torch.randn(1024, device="cuda")
store = FeatureStore(rows)
for _ in range(3):
    store.search_device(q, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 20
e0.record()
for _ in range(iters):
    store.search_device(q, k)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"scan+topk {ms:.4f} ms/query  {n*4096/ms/1e6:.1f} GB/s  ({n*4096/ms/1e6/8000*100:.1f}% of 8 TB/s)")
