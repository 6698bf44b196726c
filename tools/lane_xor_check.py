"""GPU check of the crossbar-free lane exchanges (hmm_common.h lane_xor_b32): the scan's similarity bits and the tournament's order
depend on them; the quickest witness is a scan against torch on a small store plus the 64-key sort inside it.  (The full suites --
tests/test_gpu_scan*.py, test_gpu_ops.py LayerNorm parity -- are the real test; this is the ten-second version.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippomm_amd.vector_ops import FeatureStore
g = torch.Generator(device="cuda").manual_seed(1)
rows = torch.randn(50000, 1024, generator=g, device="cuda")
q = torch.randn(1024, generator=g, device="cuda")
idx, sims = FeatureStore(rows).search_device(q, 32)
want = torch.topk((rows.double() @ q.double()) / (rows.double().norm(dim=1) * q.double().norm()), 32)
print("indices equal:", torch.equal(idx, want.indices), " max |sim diff|:", float((sims.double() - want.values).abs().max()))
assert torch.equal(idx, want.indices) and float((sims.double() - want.values).abs().max()) < 2e-6
