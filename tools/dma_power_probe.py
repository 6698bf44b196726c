"""Board power, clock and bandwidth of the global -> LDS staging stream alone (tools/csrc/dma_stream.hip), by where the data
is served from: L2 (2 MiB per XCD), Infinity Cache (16 MiB per XCD), HBM (256 MiB per XCD).  Beside it, from the same
sensor: idle, and the full fc1 GEMM.  usage: dma_power_probe.py [json_out]"""
import ctypes as C
import glob
import json
import sys
import threading
import time

from probe_common import load_probe

import torch

L, lib = load_probe()
lib.hmm_probe_dma_stream.restype = C.c_int
lib.hmm_probe_dma_stream.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_void_p, C.c_void_p]


def power_files():
    from probe_common import own_power_file                   # this GPU's sensor only (the host's sysfs lists all eight cards)
    f = own_power_file()
    return [f] if f else []


def measure(run, secs=1.5):
    """board W (average of the sensor while `run` loops), ms per run."""
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    watts, stop = [], threading.Event()

    def sample():
        files = power_files()
        while not stop.is_set():
            for p in files:
                try:
                    watts.append(int(open(p).read()) / 1e6)
                except OSError:
                    pass
            time.sleep(0.02)
    def batch():                                              # a few launches, then wait: the queue never runs ahead of the clock
        for _ in range(4):
            run()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.7:                     # let the power controller settle
        batch()
    th = threading.Thread(target=sample); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    e0.record()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        batch(); n += 4
    e1.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    return (sum(watts) / len(watts) if watts else None), e0.elapsed_time(e1) / n


ticks = torch.zeros(512, dtype=torch.int64, device="cuda")
src = torch.empty(8 * 256 * 2 ** 20, dtype=torch.uint8, device="cuda").random_()        # 2 GiB of random bytes
res = []
time.sleep(1.0)
idle = [int(open(p).read()) / 1e6 for p in power_files() for _ in range(10)]
res.append({"what": "idle", "board_W": round(sum(idle) / len(idle), 0) if idle else None})
print(res[-1], flush=True)
ITERS = 4000
for name, region in (("L2 (2 MiB per XCD)", 2 << 20), ("Infinity Cache (16 MiB per XCD)", 16 << 20), ("HBM (256 MiB per XCD)", 256 << 20)):
    run = lambda: L.check(lib.hmm_probe_dma_stream(src.data_ptr(), region, ITERS, ticks.data_ptr(), L.stream_ptr()), "dma")
    w, ms = measure(run)
    t = ticks.cpu().view(256, 2).double()
    mhz = float((t[:, 0] / t[:, 1] * 100.0).median())
    tbs = 256 * ITERS * 65536 / ms / 1e9
    rec = {"what": "global_load_lds stream served from " + name, "TB_per_s": round(tbs, 2), "board_W": round(w, 0) if w else None,
           "shader_clock_MHz": round(mhz, 0), "ms_per_launch": round(ms, 3)}
    if w and res[0]["board_W"]:
        rec["pJ_per_byte_above_idle"] = round((w - res[0]["board_W"]) / tbs, 1)
    res.append(rec)
    print(rec, flush=True)
# the real thing, same sensor
M, N, K = 65792, 5120, 1280
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
wgt = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
bias = torch.zeros(N, device="cuda")
c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
run = lambda: L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), wgt.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, 1, L.stream_ptr()), "gemm")
w, ms = measure(run, 2.0)
res.append({"what": "fc1 + GELU GEMM (M 65792, N 5120, K 1280)", "TFLOPs": round(2.0 * M * N * K / ms / 1e9, 1), "board_W": round(w, 0) if w else None,
            "LDS_fill_TB_per_s": round((M / 256) * (N / 256) * (K / 64) * 65536 / ms / 1e9, 2), "ms_per_launch": round(ms, 3)})
print(res[-1], flush=True)
x = torch.randn(M, K, device="cuda"); y = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
g, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
run = lambda: L.check(lib.hmm_op_layernorm_bf16(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, 1e-6, L.stream_ptr()), "ln")
w, ms = measure(run, 1.5)
res.append({"what": "LayerNorm 65792 x 1280 (HBM stream: 4 B read + 2 B written per element)", "TB_per_s": round(M * K * 6 / ms / 1e9, 2),
            "board_W": round(w, 0) if w else None, "ms_per_launch": round(ms, 4)})
print(res[-1], flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
