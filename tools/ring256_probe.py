"""Round 5: the 256 x 128 ring tile (eight waves of 64 x 64, three stages) against the double-buffered 128 x 128 kernel, the
eight-wave 128 x 128 ring and the ping-pong kernel on the vision GEMMs at 10-20 frames' rows, COLD weights, bit equality.
usage: ring256_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, event_ms

L, lib = load_probe()
EPI = {"bias": 0, "gelu": 1, "resid": 2}
TILES = {"db128": 0, "r128x128w8": 14, "r256x128w8": 16, "r256x128w8_stag": 17, "r256x128_2x4_stag": 18, "r256x128_2x4": 19, "pp": 3}
rows = []
st = L.stream_ptr()
MS = (3341, 4112)
for name, N, K, epi in (("vision fc2", 1280, 5120, "resid"), ("vision out", 1280, 1280, "resid"), ("vision qkv", 3840, 1280, "bias"),
                        ("vision fc1", 5120, 1280, "gelu")):
    copies = max(4, int(400e6 // (N * K * 2)) + 1)
    g = torch.Generator(device="cuda").manual_seed(N + K)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(copies)]
    bias = torch.randn(N, device="cuda", generator=g)
    for M in MS:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        c0 = torch.randn(M, N, device="cuda", generator=g)
        c = c0.clone() if epi == "resid" else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        rec = {"gemm": name, "M": M, "N": N, "K": K}
        state = {"i": 0}
        ref = None
        for rep in range(2):                     # second pass = steady clocks
            for tag, tile in TILES.items():
                def call():
                    w = ws[state["i"] % copies]
                    state["i"] += 1
                    L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, EPI[epi], tile, st), "gemm")
                rec["us_" + tag] = round(event_ms(call, 3 * copies, warmup=copies) * 1e3, 2)
                if epi == "resid":
                    c.copy_(c0)
                state["i"] = 0
                call()
                torch.cuda.synchronize()
                if ref is None:
                    ref = c.clone()
                elif not torch.equal(c, ref):
                    rec["DIFFERENT_BITS_" + tag] = True
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        if len(sys.argv) > 1:
            json.dump(rows, open(sys.argv[1], "w"), indent=1)
    del ws
