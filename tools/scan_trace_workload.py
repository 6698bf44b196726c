"""Workload for `rocprofv3 --kernel-trace`: back-to-back feature_search queries on a resident 1M x 1024 store -- 60 through
hmm_cosine_topk, 60 through hmm_cosine_topk_prefilter, 60 through the 16-query pass -- and 60 launches of the exact scan's streaming kernel alone -- each group after its own warm-up.  Also prints the HIP-event time per query of every group (under the tracer).
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/scan_trace_workload.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hippomm_amd import _lib as L
from hippomm_amd.vector_ops import FeatureStore

N, K = 1_000_000, 32
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(N, 1024, device="cuda")
for s in range(0, N, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
q16 = torch.randn(16, 1024, generator=torch.Generator(device="cuda").manual_seed(44), device="cuda")
store = FeatureStore(rows)
store.build_shadow()
out = {}
lib = L.load()
cand = torch.empty(2048 * K, dtype=torch.int64, device="cuda")
for tag, fn in (("exact", lambda: store.search_device(q, K)), ("prefilter", lambda: store.search_prefiltered_device(q, K)),
                ("multi16", lambda: store.search_multi_device(q16, K)),
                ("scan_kernel_only", lambda: L.check(lib.hmm_op_scan_topk_only(rows.data_ptr(), N, q.data_ptr(), K, cand.data_ptr(), L.stream_ptr()), "scan"))):
    for _ in range(150):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        fn()
    e1.record()
    torch.cuda.synchronize()
    out[tag + "_ms_per_query_events"] = round(e0.elapsed_time(e1) / 60, 4)
print(json.dumps(out))
