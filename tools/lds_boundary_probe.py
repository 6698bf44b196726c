"""Does a kernel that asks for a lot of LDS cost more at the kernel boundary?  Chains of 400 dependent launches of a do-nothing kernel
alternating two dynamic-LDS sizes; us per launch (HIP events).  usage: lds_boundary_probe.py [out.json]"""
import ctypes as C
import json
import sys

from probe_common import load_probe, event_ms

L, lib = load_probe()
lib.hmm_probe_empty_launches_lds.restype = C.c_int
lib.hmm_probe_empty_launches_lds.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
rows = []
K = 1024
for blocks in (64, 256, 512):
    for a, b in ((0, 0), (16 * K, 16 * K), (64 * K, 64 * K), (65 * K, 65 * K), (96 * K, 96 * K), (128 * K, 128 * K), (160 * K, 160 * K),
                 (0, 64 * K), (0, 96 * K), (0, 128 * K), (64 * K, 128 * K), (96 * K, 128 * K)):
        ms = event_ms(lambda: L.check(lib.hmm_probe_empty_launches_lds(400, blocks, 256, a, b, L.stream_ptr()), "empty"), 5, warmup=2)
        rec = {"blocks": blocks, "lds_a_kib": a // K, "lds_b_kib": b // K, "us_per_launch": round(ms * 1e3 / 400, 3)}
        rows.append(rec)
        print(rec, flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
