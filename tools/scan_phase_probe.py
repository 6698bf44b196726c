"""Where do the scan kernel's microseconds go?  Pure streaming read vs dot products only (hmm_op_scan_sims) vs dot products +
per-block candidate lists (hmm_op_scan_topk_only) vs the whole query (hmm_cosine_topk), same 1M x 1024 store."""
import ctypes as C
import json
import sys

from probe_common import load_probe, event_ms

import torch

L, lib = load_probe()
lib.hmm_probe_hbm_read.restype = C.c_int
lib.hmm_probe_hbm_read.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
N, D, K = 1_000_000, 1024, 32
rows = torch.empty(N, D, device="cuda").normal_()
q = torch.randn(D, device="cuda")
sims = torch.empty(N, device="cuda")
sink = torch.zeros(4, device="cuda")
cand = torch.zeros(4096 * 128, dtype=torch.int64, device="cuda")
ws_bytes = lib.hmm_cosine_topk_workspace_bytes(N, K)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
idx = torch.empty(K, dtype=torch.int64, device="cuda"); sim = torch.empty(K, device="cuda"); n_out = torch.zeros(1, dtype=torch.int32, device="cuda")
S = L.stream_ptr
res = {}
res["pure_read_ms"] = event_ms(lambda: L.check(lib.hmm_probe_hbm_read(rows.data_ptr(), N * D * 4, 2048, 8, 1, sink.data_ptr(), S()), "r"), 20, warmup=3)
res["scan_sims_ms"] = event_ms(lambda: L.check(lib.hmm_op_scan_sims(rows.data_ptr(), N, q.data_ptr(), sims.data_ptr(), S()), "s"), 20, warmup=3)
res["scan_topk_only_ms"] = event_ms(lambda: L.check(lib.hmm_op_scan_topk_only(rows.data_ptr(), N, q.data_ptr(), K, cand.data_ptr(), S()), "t"), 20, warmup=3)
res["whole_query_ms"] = event_ms(lambda: L.check(lib.hmm_cosine_topk(rows.data_ptr(), N, D, q.data_ptr(), K, idx.data_ptr(), sim.data_ptr(), n_out.data_ptr(),
                                                                     ws.data_ptr(), ws_bytes, S()), "q"), 20, warmup=3)
res = {k: round(v, 4) for k, v in res.items()}
res["GBps"] = {k[:-3]: round(N * D * 4 / v / 1e6, 1) for k, v in res.items()}
print(res)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
