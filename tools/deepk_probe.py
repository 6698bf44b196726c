"""Few-row GEMMs with COLD weights (each call takes the next of enough weight copies to exceed the 256-MiB Infinity Cache, as in a
tower, where a block's weights are read once per forward): the ring kernels against their deep-K variants (2 / 4 K-tiles per
ring stage) and the sliver kernel, and bit equality with the double-buffered 128x128 kernel.  usage: deepk_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, event_ms

L, lib = load_probe()
EPI = {"bias": 0, "gelu": 1, "resid": 2}
TILES = {"128x128": 0, "ring128": 6, "ring64": 7, "ring32": 8, "ring64_k2": 9, "ring64_k4": 10, "ring32_k2": 11, "ring32_k4": 12,
         "sliver": 5, "auto": -1}
rows = []


def run(M, N, K, epi, tile, copies):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(copies)]
    bias = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(M, N, device="cuda", generator=g)
    c = c0.clone() if epi == "resid" else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    st = L.stream_ptr()
    state = {"i": 0}

    def call():
        w = ws[state["i"] % copies]
        state["i"] += 1
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, EPI[epi], tile, st), "gemm")
    ms = event_ms(call, 3 * copies, warmup=copies)
    if epi == "resid":
        c.copy_(c0)
    state["i"] = 0
    call()
    torch.cuda.synchronize()
    return ms * 1e3, c.clone()


SHAPES = [("text qkv", 3072, 1024, "bias"), ("text out", 1024, 1024, "resid"), ("text fc1", 4096, 1024, "gelu"),
          ("text fc2", 1024, 4096, "resid"), ("vision qkv", 3840, 1280, "bias"), ("vision out", 1280, 1280, "resid"),
          ("vision fc1", 5120, 1280, "gelu"), ("vision fc2", 1280, 5120, "resid"), ("audio qkv", 2304, 768, "bias"),
          ("audio fc1", 3072, 768, "gelu"), ("audio fc2", 768, 3072, "resid")]
for name, N, K, epi in SHAPES:
    copies = max(4, int(400e6 // (N * K * 2)) + 1)
    Ms = (77, 154, 308) if name.startswith("text") else ((229 * 3, 229 * 6) if name.startswith("audio") else (16, 257, 514, 1028))
    for M in Ms:
        rec = {"gemm": name, "M": M, "N": N, "K": K, "weight_copies": copies}
        ref = None
        for tag, tile in TILES.items():
            if tag.endswith("_k4") and (K // 64) % 4:
                continue
            if tag == "ring32" or tag.startswith("ring32_"):
                if ((M + 31) // 32) * (N // 32) > 2048:
                    continue
            if tag == "sliver" and M > 320:
                continue
            us, c = run(M, N, K, epi, tile, copies)
            if ref is None:
                ref = c
            rec["us_" + tag] = round(us, 2)
            if not torch.equal(c, ref):
                rec["DIFFERENT_BITS_" + tag] = True
        rows.append(rec)
        print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
