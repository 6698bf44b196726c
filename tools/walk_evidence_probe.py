"""Tile walks of the ping-pong GEMM (probe build, knob g_gemm_walk) on the fc1 / fc2 / qkv shapes at M = 65 536: per walk the
kernel time (plain launch, interleaved rounds), and from in-kernel stamps the tile time, the fill + main-loop time and the
shader clock held inside fill + main loop.  FETCH_SIZE per walk comes from the same launches under
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/walk_evidence_probe.py --pmc
(summarised by tools/walk_pmc_summarize.py).  usage: walk_evidence_probe.py [--pmc] [json_out]"""
import ctypes as C
import json
import sys

from probe_common import load_probe, setter, event_ms

import numpy as np
import torch

L, lib = load_probe()
lib.hmm_probe_set_gemm_stamps.restype = None
lib.hmm_probe_set_gemm_stamps.argtypes = [C.c_void_p]
pmc = "--pmc" in sys.argv
out_path = [a for a in sys.argv[1:] if not a.startswith("--")]
M = 65536
WALKS = [(0, "strips"), ((8 << 8) | 4, "8x4"), ((4 << 8) | 8, "4x8"), ((16 << 8) | 2, "16x2"), ((6 << 8) | 5, "6x5"), ((3 << 8) | 10, "3x10")]
shapes = [("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2), ("qkv", 3840, 1280, 0)]
res = {}
for name, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    n_wg = (M // 256) * (N // 256)
    stamps = torch.zeros(n_wg * 8, dtype=torch.int64, device="cuda")
    run = lambda: L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, 3,
                                                    L.stream_ptr()), "gemm")
    if pmc:                                            # fixed launch count per walk: 2 warm + 4 counted, in WALKS order
        for walk, tag in WALKS:
            if N // 256 < (walk & 255):
                continue
            setter(lib, "g_gemm_walk")(walk)
            for _ in range(6):
                run()
            torch.cuda.synchronize()
            print(f"pmc {name} {tag} done", flush=True)
        continue
    times = {tag: [] for _, tag in WALKS}
    for rnd in range(3):
        for walk, tag in WALKS:
            if N // 256 < (walk & 255):
                continue
            setter(lib, "g_gemm_walk")(walk)
            times[tag].append(event_ms(run, 8, warmup=2))
    for walk, tag in WALKS:
        if not times[tag]:
            continue
        setter(lib, "g_gemm_walk")(walk)
        for _ in range(3):
            run()
        lib.hmm_probe_set_gemm_stamps(stamps.data_ptr())
        run(); torch.cuda.synchronize()
        lib.hmm_probe_set_gemm_stamps(None)
        s = stamps.cpu().numpy().reshape(n_wg, 8)
        loop_us = (s[:, 2] - s[:, 0]) / 100.0
        tile_us = (s[:, 3] - s[:, 0]) / 100.0
        clk = (s[:, 7] - s[:, 6]) / ((s[:, 2] - s[:, 0]) * 10.0)
        ms = sorted(times[tag])[1]
        rec = {"kernel_us": round(ms * 1e3, 1), "tflops": round(2.0 * M * N * K / ms / 1e9, 1),
               "fill_plus_mainloop_us_p50": round(float(np.median(loop_us)), 2), "tile_us_p50": round(float(np.median(tile_us)), 2),
               "in_kernel_clock_GHz_p10_p50_p90": [round(float(np.percentile(clk, q)), 3) for q in (10, 50, 90)]}
        res[f"{name}_{tag}"] = rec
        print(name, tag, rec, flush=True)
    del a, w, c
setter(lib, "g_gemm_walk")(0)
if out_path and not pmc:
    json.dump(res, open(out_path[0], "w"), indent=1)
