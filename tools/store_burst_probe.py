"""How fast do the CUs retire a GEMM epilogue's output burst, as a function of how many CUs per XCD and how many XCDs
store at the same time?  (tools/csrc/store_burst.hip)   usage: store_burst_probe.py [json_out]"""
import ctypes as C
import json
import sys

from probe_common import load_probe

import torch

L, lib = load_probe()
lib.hmm_probe_store_burst.restype = C.c_int
lib.hmm_probe_store_burst.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

N_COLS, TILES_N, REPS = 3840, 15, 8
rows = 256 * ((256 * REPS + TILES_N - 1) // TILES_N + 1)
ticks = torch.zeros(512, dtype=torch.int64, device="cuda")
results = []
for mode, name, esz, tile_bytes in ((0, "bf16 store", 2, 131072), (2, "fp32 store", 4, 262144), (1, "fp32 rmw", 4, 524288)):
    out = torch.zeros(rows, N_COLS * esz // 2, dtype=torch.int16, device="cuda")   # rows x n_cols elements of esz bytes
    for n_xcd in (1, 2, 4, 8):
        for per_xcd in (1, 2, 4, 8, 16, 32):
            def run():
                L.check(lib.hmm_probe_store_burst(out.data_ptr(), N_COLS, TILES_N, n_xcd, per_xcd, REPS, mode,
                                                  ticks.data_ptr(), L.stream_ptr()), "burst")
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            spans = []
            for _ in range(5):
                ticks.zero_()
                run()
                torch.cuda.synchronize()
                t = ticks.cpu().view(256, 2)
                act = t[:, 1] > 0
                d = (t[act, 1] - t[act, 0]).double() * 10.0 / 1e3          # 100 MHz ticks -> us
                spans.append((float(d.median()), float(d.max()), float((t[act, 1].max() - t[act, 0].min()) * 10.0 / 1e3)))
            spans.sort()
            med, worst, wall = spans[len(spans) // 2]
            active = n_xcd * per_xcd
            rec = {"mode": name, "n_xcd": n_xcd, "per_xcd": per_xcd, "us_per_tile_median_wg": round(med / REPS, 3),
                   "us_per_tile_slowest_wg": round(worst / REPS, 3),
                   "GBps_per_wg": round(tile_bytes * REPS / med / 1e3, 1),
                   "TBps_chip": round(tile_bytes * REPS * active / wall / 1e6, 3)}
            results.append(rec)
            print(rec, flush=True)
    del out
if len(sys.argv) > 1:
    json.dump(results, open(sys.argv[1], "w"), indent=1)
