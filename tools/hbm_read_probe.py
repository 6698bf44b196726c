"""Read bandwidth a pure streaming kernel gets from HBM (tools/csrc/hbm_read.hip): the practical ceiling for the scan.
usage: hbm_read_probe.py [json_out]"""
import ctypes as C
import json
import sys

from probe_common import load_probe, event_ms

import torch

L, lib = load_probe()
lib.hmm_probe_hbm_read.restype = C.c_int
lib.hmm_probe_hbm_read.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
N_BYTES = 4_096_000_000                                   # the scan's store: 1M x 1024 fp32
src = torch.empty(N_BYTES // 4, dtype=torch.float32, device="cuda").normal_()
sink = torch.zeros(4, device="cuda")
res = []
for nt in (1, 0):
    for unroll in (4, 8, 16):
        for blocks in (1024, 2048, 4096, 8192):
            ms = event_ms(lambda: L.check(lib.hmm_probe_hbm_read(src.data_ptr(), N_BYTES, blocks, unroll, nt, sink.data_ptr(),
                                                                  L.stream_ptr()), "read"), 10, warmup=3)
            rec = {"nt": nt, "loads_in_flight_per_lane": unroll, "blocks": blocks, "ms": round(ms, 4),
                   "GBps": round(N_BYTES / ms / 1e6, 1), "frac_of_8TBps": round(N_BYTES / ms / 1e6 / 8000, 4)}
            res.append(rec)
            print(rec, flush=True)
# reference: a device-to-device copy of the same bytes (read + write)
dst = torch.empty_like(src)
ms = event_ms(lambda: dst.copy_(src), 10, warmup=3)
print({"torch_copy_ms": round(ms, 4), "read_plus_write_GBps": round(2 * N_BYTES / ms / 1e6, 1)})
res.append({"torch_copy_ms": round(ms, 4), "read_plus_write_GBps": round(2 * N_BYTES / ms / 1e6, 1)})
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
