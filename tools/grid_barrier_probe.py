"""What a grid-wide barrier costs next to a dependent kernel launch (tools/csrc/grid_barrier.hip).  usage: grid_barrier_probe.py [json_out]"""
import ctypes as C
import json
import sys

from probe_common import load_probe, event_ms

import torch

L, lib = load_probe()
lib.hmm_probe_grid_barrier.restype = C.c_int
lib.hmm_probe_grid_barrier.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
lib.hmm_probe_empty_launches.restype = C.c_int
lib.hmm_probe_empty_launches.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
ctr = torch.zeros(512, dtype=torch.int32, device="cuda")
buf = torch.zeros(256 * 65536, dtype=torch.float32, device="cuda")
stamps = torch.zeros(512, dtype=torch.int64, device="cuda")
res = []
ROUNDS = 1000
for threads in (256,):
    for blocks in (64, 128, 256):
        for mode in (0, 3, 4, 5):
            for work in (0, 4096):
                def run():
                    L.check(lib.hmm_probe_grid_barrier(ctr.data_ptr(), blocks, threads, ROUNDS, mode, buf.data_ptr(), work,
                                                       stamps.data_ptr(), L.stream_ptr()), "grid_barrier")
                ms = event_ms(run, 3, warmup=1)
                s = stamps.cpu()[: 2 * blocks].view(-1, 2)
                inside = float((s[:, 1].max() - s[:, 0].min()).item()) / 100.0 / ROUNDS        # s_memrealtime: 100 MHz
                rec = {"threads": threads, "blocks": blocks, "mode": mode, "work_floats_per_wg": work,
                       "us_per_round_events": round(ms * 1e3 / ROUNDS, 3), "us_per_round_in_kernel": round(inside, 3),
                       "spin_limit_hit": int(ctr[15].item())}
                res.append(rec)
                print(rec, flush=True)
for blocks, threads in ((1, 64), (256, 256), (1024, 256)):
    for n in (200,):
        ms = event_ms(lambda: L.check(lib.hmm_probe_empty_launches(n, blocks, threads, L.stream_ptr()), "empty"), 5, warmup=2)
        rec = {"empty_launches": n, "blocks": blocks, "threads": threads, "us_per_launch": round(ms * 1e3 / n, 3)}
        res.append(rec)
        print(rec, flush=True)
# the same chain replayed as a HIP graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    L.check(lib.hmm_probe_empty_launches(200, 256, 256, C.c_void_p(s.cuda_stream)), "empty")
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        L.check(lib.hmm_probe_empty_launches(200, 256, 256, C.c_void_p(s.cuda_stream)), "empty")
ms = event_ms(lambda: g.replay(), 5, warmup=2)
rec = {"graph_replay_of_empty_launches": 200, "us_per_node": round(ms * 1e3 / 200, 3)}
res.append(rec)
print(rec, flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
