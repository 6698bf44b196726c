"""Round 5's build of the library (tools/libhippomm_r5.so, see scan_ab_r5_probe.py) against this round's on whole forwards, interleaved in
one process: vision 1 / 6 / 32 / 256 frames, one audio segment, one question.  Guards the numbers no GPU-minute was meant to move.
    python tools/forward_ab_r5_probe.py [out.json]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from hippomm_amd import _lib as L
from hippomm_amd.encoder import HipTower, synthetic_state_dict


def bind_subset(path):
    lib = C.CDLL(path)
    for name, (res, args) in L._SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return lib


libs = {"round6": L.load(), "round5": bind_subset(os.path.join(ROOT, "tools", "libhippomm_r5.so"))}
cases = [("vision", 1), ("vision", 6), ("vision", 32), ("vision", 256), ("audio", 1), ("text", 1)]
towers = {}
for tag, lib in libs.items():
    L._lib = lib                                     # HipTower binds whatever _lib.load() returns at construction
    for kind in ("vision", "audio", "text"):
        towers[(tag, kind)] = HipTower(kind, synthetic_state_dict((kind,), seed=1234))
L._lib = libs["round6"]
out = {}
gen = torch.Generator(device="cuda").manual_seed(5)
for kind, b in cases:
    if kind == "vision":
        x = torch.randn(b, 3, 224, 224, device="cuda", generator=gen)
    elif kind == "audio":
        x = torch.randn(b, 3, 1, 128, 204, device="cuda", generator=gen)
    else:
        x = torch.randint(1, 49000, (b, 77), device="cuda", generator=gen)
        x[:, 0], x[:, 20] = 49406, 49407
    emb = {t: torch.empty(b, 1024, device="cuda") for t in libs}
    iters = 8 if b >= 256 else 40
    ms = {t: [] for t in libs}
    for rep in range(4):
        for t in libs:
            tw = towers[(t, kind)]
            for _ in range(3):
                tw.forward_into(x, emb[t])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                tw.forward_into(x, emb[t])
            torch.cuda.synchronize()
            ms[t].append((time.perf_counter() - t0) / iters * 1e3)
    out[f"{kind}_{b}"] = {"round5_ms": round(min(ms["round5"]), 4), "round6_ms": round(min(ms["round6"]), 4),
                          "ratio_r6_over_r5": round(min(ms["round6"]) / min(ms["round5"]), 4),
                          "same_bits": bool(torch.equal(emb["round5"], emb["round6"]))}
    print(f"{kind}_{b}", json.dumps(out[f"{kind}_{b}"]), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
