"""A/B of ONE probe-build knob on whole forwards: wall clock per forward at the reference's call sizes and around them, the two
settings interleaved three times, bit equality of the embeddings.  usage: knob_ab_probe.py <knob> <value_a> <value_b> [out.json]
e.g.  knob_ab_probe.py g_gemm_ring8 0 1"""
import json
import sys
import time

import torch

from probe_common import load_probe, setter

L, lib = load_probe()
knob, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out_path = sys.argv[4] if len(sys.argv) > 4 else None
only = sys.argv[5] if len(sys.argv) > 5 else None                 # e.g. vision:32,36,40 -- one tower, these batch sizes
set_knob = setter(lib, knob)
from hippomm_amd.encoder import HipTower, synthetic_state_dict   # noqa: E402


def wall_ms(fn, iters):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


rows = []
plan = (("text", (1, 4, 9, 16, 32, 64, 96)), ("vision", (1, 2, 3, 8, 12, 13, 14, 15, 16, 20, 22, 24, 26, 28, 32, 48, 64, 256)),
        ("audio", (1, 2, 4, 6, 8, 12, 16, 32, 128)))
if only:
    plan = ((only.split(":")[0], tuple(int(b) for b in only.split(":")[1].split(","))),)
for kind, batches in plan:
    tower = HipTower(kind, synthetic_state_dict((kind,), seed=99))
    for B in batches:
        if kind == "text":
            x = torch.randint(1, 49000, (B, 77), device="cuda")
            x[:, 0], x[:, 20] = 49406, 49407
        elif kind == "audio":
            x = torch.randn(B, 3, 1, 128, 204, device="cuda")
        else:
            x = torch.randn(B, 3, 224, 224, device="cuda")
        out = torch.empty(B, 1024, device="cuda")
        iters = 30 if B <= 32 else 8
        ms = {va: [], vb: []}
        outs = {}
        for rep in range(3):
            for v in (va, vb):
                set_knob(v)
                ms[v].append(wall_ms(lambda: tower.forward_into(x, out), iters))
                outs[v] = out.clone()
        rec = {"tower": kind, "batch": B, f"ms_{knob}={va}": round(min(ms[va]), 4), f"ms_{knob}={vb}": round(min(ms[vb]), 4),
               "ratio_b_over_a": round(min(ms[vb]) / min(ms[va]), 4), "same_bits": bool(torch.equal(outs[va], outs[vb]))}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        if out_path:
            json.dump(rows, open(out_path, "w"), indent=1)
    del tower
    torch.cuda.empty_cache()
