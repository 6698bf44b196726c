"""One question through the text tower (24 blocks, 77 token rows): latency with the few-row GEMMs on the tiled kernels and on
the sliver kernel, eager and as a replayed HIP graph; the same for 2 and 4 questions and for one / eight frames through the
vision tower.  Checks that both dispatches give the same embedding bits.  Usage: python tools/text_latency_probe.py [out.json]"""
import json
import sys
import time

import torch

from probe_common import load_probe, setter

L, lib = load_probe()
set_auto = setter(lib, "g_gemm_sliver_auto")
set_stages = setter(lib, "g_gemm_small_stages")
set_64 = setter(lib, "g_gemm_small_64")
set_32 = setter(lib, "g_gemm_small_32")
set_deepk = setter(lib, "g_gemm_deepk")
set_qsplit = setter(lib, "g_attn_q_split")
set_short = setter(lib, "g_attn_short_keys")
from hippomm_amd.encoder import HipTower, synthetic_state_dict   # noqa: E402

rows = []


def wall_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def graph_of(tower, x, out):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tower.forward_into(x, out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            tower.forward_into(x, out)
    torch.cuda.synchronize()
    return g


for kind, batches in (("text", (1, 2, 3, 4, 8, 16, 64)), ("vision", (1, 2, 4, 8, 32)), ("audio", (1, 2, 4, 8))):
    sd = synthetic_state_dict((kind,), seed=99)
    tower = HipTower(kind, sd)
    del sd
    for B in batches:
        if kind == "text":
            x = torch.randint(1, 49000, (B, 77), device="cuda")
            x[:, 0], x[:, 20] = 49406, 49407
        elif kind == "audio":
            x = torch.randn(B, 3, 1, 128, 204, device="cuda")
        else:
            x = torch.randn(B, 3, 224, 224, device="cuda")
        rec = {"tower": kind, "batch": B}
        outs = []
        for tag, auto, stages, t64, t32, dk, qs in (("tiled", 0, 2, 0, 0, 0, 0), ("ring128", 0, 4, 0, 0, 0, 0), ("ring64", 0, 4, 512, 0, 0, 0),
                                                    ("ring64_sliver", 1, 4, 512, 0, 0, 0), ("round3_product", 1, 4, 512, 400, 0, 0),
                                                    ("deepk", 1, 4, 512, 400, 1, 0), ("product", 1, 4, 512, 400, 1, 1)):
            set_deepk(dk)
            set_qsplit(qs)
            set_short(1 if tag == "product" else 0)
            set_auto(auto)
            set_stages(stages)
            set_64(t64)
            set_32(t32)
            out = torch.empty(B, 1024, device="cuda")
            rec[f"ms_eager_{tag}"] = round(wall_ms(lambda: tower.forward_into(x, out)), 3)
            outs.append(out.clone())
        rec["same_bits"] = all(bool(torch.equal(outs[0], o)) for o in outs[1:])
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    del tower
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
