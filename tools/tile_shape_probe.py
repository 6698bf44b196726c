"""Four waves with 128 x 128 accumulator tiles against eight waves with 128 x 64 tiles, with and without each wave's share of the
global -> LDS staging stream (tools/csrc/mfma_tile_shape.hip): TFLOP/s, in-kernel clock, board power.
usage: tile_shape_probe.py [json_out]"""
import ctypes as C
import json
import sys
import threading
import time

from probe_common import load_probe, own_power_file

import torch

L, lib = load_probe()
lib.hmm_probe_mfma_tile_shape.restype = C.c_int
lib.hmm_probe_mfma_tile_shape.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
STEPS = 100_000
FLOPS = 256 * STEPS * 2 * 256 * 256 * 32                      # per launch: a 256 x 256 x 32 slice per CU and step
src = torch.empty(16 << 20, dtype=torch.uint8, device="cuda").random_()
sink = torch.zeros(512, device="cuda")
ticks = torch.zeros(512, dtype=torch.int64, device="cuda")
pf = own_power_file()
res = []
for rnd in range(1):
    for waves, dma in ((8, 0), (4, 0), (8, 1), (4, 1), (8, 2), (4, 2), (8, 3), (4, 3)):
        run = lambda: L.check(lib.hmm_probe_mfma_tile_shape(waves, dma, STEPS, src.data_ptr(), sink.data_ptr(), ticks.data_ptr(), L.stream_ptr()), "tile_shape")
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 2.0:                 # the power sensor settles over a second or two
            run(); run()
            torch.cuda.synchronize()
        watts, stop = [], threading.Event()

        def sample():
            while not stop.is_set():
                if pf:
                    try:
                        watts.append(int(open(pf).read()) / 1e6)
                    except OSError:
                        pass
                time.sleep(0.02)
        th = threading.Thread(target=sample); th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(24):
            run()
        e1.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        ms = e0.elapsed_time(e1) / 24
        t = ticks.cpu().view(256, 2).double()
        rec = {"waves": waves, "wave_tile": "128x64" if waves == 8 else "128x128", "with_lds_fill_stream": bool(dma), "steps_of_fill_in_flight": dma,
               "TFLOPs": round(FLOPS / ms / 1e9, 1), "shader_clock_MHz": round(float((t[:, 0] / t[:, 1] * 100).median()), 0),
               "board_W": round(sum(watts) / len(watts), 0) if watts else None, "ms_per_launch": round(ms, 2),
               "lds_read_TB_per_s": round(256 * STEPS * waves * (12 if waves == 8 else 16) * 1024 / ms / 1e9, 1),
               "lds_fill_TB_per_s": round(256 * STEPS * 32768 / ms / 1e9, 2) if dma else 0.0}
        res.append(rec)
        print(rec, flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
