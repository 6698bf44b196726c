"""Per-event scan throughput: 1M rows in 2000 events of 500 frames, k=5 (the reference's per-event k)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from hippomm_amd.vector_ops import FeatureStore
n, E, k = 1_000_000, 2000, 5
rows = torch.randn(n, 1024, device="cuda")
q = torch.randn(1024, device="cuda")
fs = FeatureStore(rows)
off = torch.arange(0, n + 1, n // E, dtype=torch.int64, device="cuda")
for _ in range(3): fs.search_segments_device(q, off, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fs.search_segments_device(q, off, k)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"segmented scan: {ms:.4f} ms for {E} events ({ms*1e3/E:.2f} us/event)  {n*4096/ms/1e6:.0f} GB/s")
# the reference's structure: one scan call per event
per = n // E
stores = [FeatureStore(rows[i * per:(i + 1) * per]) for i in range(200)]
for s in stores[:5]: s.search_device(q, k)
torch.cuda.synchronize()
e0.record()
for s in stores: s.search_device(q, k)
e1.record(); torch.cuda.synchronize()
print(f"per-event launches: {e0.elapsed_time(e1)/200*1e3:.1f} us/event")
