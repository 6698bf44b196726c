"""Sustained bf16 MFMA rate of the whole chip under its power cap, by instruction shape and with / without the LDS operand
traffic of the real GEMM (tools/csrc/mfma_power.hip).   usage: mfma_power_probe.py [json_out]"""
import ctypes as C
import glob
import json
import sys
import threading
import time

from probe_common import load_probe

import torch

L, lib = load_probe()
lib.hmm_probe_mfma_power.restype = C.c_int
lib.hmm_probe_mfma_power.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

NAMES = {0: "16x16x32 bf16, operands in registers", 1: "32x32x16 bf16, operands in registers",
         2: "16x16x32 bf16 + 12 ds_read_b128 per K=32 step", 3: "32x32x16 bf16 + 12 ds_read_b128 per K=32 step",
         4: "16x16x32 bf16, operands in registers, CHANGING every step (xor masks)",
         6: "16x16x32 bf16 + 12 ds_read_b128 per K=32 step of a DIFFERENT fragment set every step"}
STEPS = 200_000                                             # ~50 ms per launch at 1.5 GHz
FLOPS = 256 * 8 * STEPS * 2 * 128 * 64 * 32                 # per launch


def power_files():
    """Only THIS GPU's sensor (round 2 averaged over every card of the host: its ~400 W readings were 1/8 load + 7/8 idle)."""
    from probe_common import own_power_file
    f = own_power_file()
    return [f] if f else []


def sample_power(stop, out):
    files = power_files()
    while not stop.is_set():
        for p in files:
            try:
                out.append(int(open(p).read()) / 1e6)
            except OSError:
                pass
        time.sleep(0.02)


sink = torch.zeros(512, device="cuda")
ticks = torch.zeros(512, dtype=torch.int64, device="cuda")
results = []
for mode in (0, 4, 2, 6, 1, 0, 4):
    def run():
        L.check(lib.hmm_probe_mfma_power(mode, STEPS, sink.data_ptr(), ticks.data_ptr(), L.stream_ptr()), "mfma_power")
    for _ in range(20):                                     # ~1 s: let the power controller settle
        run()
    torch.cuda.synchronize()
    watts, stop = [], threading.Event()
    th = threading.Thread(target=sample_power, args=(stop, watts))
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / 20
    t = ticks.cpu().view(256, 2).double()
    mhz = (t[:, 0] / t[:, 1] * 100.0)
    rec = {"mode": NAMES[mode], "ms_per_launch": round(ms, 3), "TFLOPs": round(FLOPS / ms / 1e9, 1),
           "frac_of_2500": round(FLOPS / ms / 1e9 / 2500, 3),
           "shader_clock_MHz_median": round(float(mhz.median()), 0), "shader_clock_MHz_min": round(float(mhz.min()), 0),
           "board_W_avg": round(sum(watts) / len(watts), 0) if watts else None}
    results.append(rec)
    print(rec, flush=True)
if len(sys.argv) > 1:
    json.dump(results, open(sys.argv[1], "w"), indent=1)
