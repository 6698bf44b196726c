"""Where does a 256x256 GEMM tile spend its time?  In-kernel stamps of the probe build (tools/libhippomm_probe.so):
per workgroup start / first K-tile landed / main loop done / stores retired (s_memrealtime, 10 ns), XCC and HW ids.
usage: gemm_stamp_probe.py [json_out]"""
import ctypes as C
import json
import sys

from probe_common import load_probe

import numpy as np
import torch

L, lib = load_probe()
lib.hmm_probe_set_gemm_stamps.restype = None
lib.hmm_probe_set_gemm_stamps.argtypes = [C.c_void_p]
M = 65536                                         # whole rounds only: the ping-pong kernel alone, no peeled tail
shapes = [("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2)]
res = {}
for name, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    n_wg = (M // 256) * (N // 256)
    stamps = torch.zeros(n_wg * 8, dtype=torch.int64, device="cuda")

    def run():
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, 3,
                                          L.stream_ptr()), "gemm")
    lib.hmm_probe_set_gemm_stamps(None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    plain_ms = e0.elapsed_time(e1)
    lib.hmm_probe_set_gemm_stamps(stamps.data_ptr())
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    stamped_ms = e0.elapsed_time(e1)
    lib.hmm_probe_set_gemm_stamps(None)
    s = stamps.cpu().numpy().reshape(n_wg, 8)
    t0 = s[:, 0].min()
    start, landed, loop_done, end = [(s[:, i] - t0) / 100.0 for i in range(4)]          # us
    landed = start                                                                      # slot 1 is no longer stamped: fill + loop together
    loop_clock_ghz = (s[:, 7] - s[:, 6]) / ((s[:, 2] - s[:, 0]) * 10.0)                 # shader ticks per ns over fill + main loop
    xcc, hw = s[:, 4] & 0xF, s[:, 5]
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)            # cu_id | sh_id | se_id
    rounds = n_wg // 256
    order = np.argsort(start)
    rec = {"M": M, "N": N, "K": K, "workgroups": n_wg, "kernel_ms_plain": round(plain_ms, 4),
           "kernel_ms_stamped": round(stamped_ms, 4),
           "fill_plus_mainloop_us": [round(float(np.percentile(loop_done - start, q)), 2) for q in (10, 50, 90)],
           "in_kernel_clock_GHz_fill_plus_mainloop": [round(float(np.percentile(loop_clock_ghz, q)), 3) for q in (10, 50, 90)],
           "epilogue_us": [round(float(np.percentile(end - loop_done, q)), 2) for q in (10, 50, 90)],
           "tile_us": [round(float(np.percentile(end - start, q)), 2) for q in (10, 50, 90)],
           "last_end_us": round(float(end.max()), 2),
           "xcc_matches_blockidx_mod8": bool((np.bincount(((s[:, 4] & 0xF) * 8 + (np.arange(n_wg) % 8)).astype(int), minlength=64).reshape(8, 8) > 0).sum() == 8),
           "distinct_cus_round0": int(len(set(zip(xcc[order[:256]].tolist(), cu[order[:256]].tolist()))))}
    # spread of the epilogue starts inside each round (by start order): are the CUs still in lock-step late in the launch?
    spread = []
    for r in range(rounds):
        idx = order[r * 256:(r + 1) * 256]
        spread.append(round(float(np.percentile(loop_done[idx], 95) - np.percentile(loop_done[idx], 5)), 2))
    rec["epilogue_start_spread_p5_p95_us_per_round"] = spread
    ep = end - loop_done
    rec["epilogue_us_by_round_median"] = [round(float(np.median(ep[order[r * 256:(r + 1) * 256]])), 2) for r in range(rounds)]
    # hand-over on one CU: end of a workgroup's stores -> first instruction of the next workgroup on the same CU
    gaps, key = [], xcc.astype(np.int64) * 4096 + cu
    for kcu in np.unique(key):
        sel = np.where(key == kcu)[0]
        sel = sel[np.argsort(start[sel])]
        gaps.extend((start[sel[1:]] - end[sel[:-1]]).tolist())
    rec["cu_handover_gap_us"] = [round(float(np.percentile(gaps, q)), 2) for q in (10, 50, 90)]
    rec["workgroups_per_cu"] = [int(x) for x in np.percentile(np.bincount(np.unique(key, return_inverse=True)[1]), (0, 50, 100))]
    res[name] = rec
    print(name, rec, flush=True)
    del a, w, c
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
