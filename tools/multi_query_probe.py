"""Batched scan: time per pass and per query at N = 1M."""
import sys
sys.path.insert(0, ".")
import torch
from hippomm_amd.vector_ops import FeatureStore
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(N, 1024, device="cuda")
for s in range(0, N, 65536):
    rows[s:s + 65536] = torch.nn.functional.normalize(torch.randn(min(65536, N - s), 1024, device="cuda", generator=g), dim=1)
fs = FeatureStore(rows)
from hippomm_amd import _lib
for rot in (1, 5):
  _lib.load().hmm_dev_set_multi_rotate(rot)
  print("flags (1 rotate, 2 no MFMA, 4 no selection):", rot)
  for nq, k in [(1, 32), (16, 32), (16, 5), (32, 32)]:
    q = torch.randn(nq, 1024, device="cuda", generator=g)
    for _ in range(3): fs.search_multi_device(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fs.search_multi_device(q, k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    passes = (nq + 15) // 16
    print(f"N={N} Q={nq} k={k}: {ms:.3f} ms  ({ms/nq*1e3:.1f} us per query; {N*4096*passes/ms/1e9:.2f} TB/s of store reads)", flush=True)
q1 = torch.randn(1024, device="cuda", generator=g)
for _ in range(3): fs.search_device(q1, 32)
torch.cuda.synchronize()
e0.record()
for _ in range(10): fs.search_device(q1, 32)
e1.record(); torch.cuda.synchronize()
print(f"single-query scan: {e0.elapsed_time(e1)/10:.3f} ms")
