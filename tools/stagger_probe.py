"""Round 5: the staggered loop in the eight-wave ring tiles (gemm_bf16.hip STG = 1) against the plain eight-wave tiles, the
four-wave 128x128 ring and the 256x256 ping-pong kernel: the vision / audio GEMMs at 1 ... 16 frames' rows, COLD weights (a new
weight copy per call), bit equality with the double-buffered 128x128 kernel.
usage: stagger_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, event_ms

L, lib = load_probe()
EPI = {"bias": 0, "gelu": 1, "resid": 2}
TILES = {"db128": 0, "ring128": 6, "r128x128w8": 14, "r128x128w8s": 16, "r128x64w8": 15, "r128x64w8s": 17, "pp": 3}
rows = []
st = L.stream_ptr()
VIS = (257, 771, 2056, 2570, 4112)
SHAPES = [("vision qkv", 3840, 1280, "bias", VIS), ("vision fc1", 5120, 1280, "gelu", VIS), ("vision out", 1280, 1280, "resid", VIS),
          ("vision fc2", 1280, 5120, "resid", VIS), ("audio fc1", 3072, 768, "gelu", (229, 687, 2061)), ("audio fc2", 768, 3072, "resid", (229, 687, 2061))]
for name, N, K, epi, Ms in SHAPES:
    copies = max(4, int(400e6 // (N * K * 2)) + 1)
    g = torch.Generator(device="cuda").manual_seed(N + K)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(copies)]
    bias = torch.randn(N, device="cuda", generator=g)
    for M in Ms:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        c0 = torch.randn(M, N, device="cuda", generator=g)
        c = c0.clone() if epi == "resid" else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        rec = {"gemm": name, "M": M, "N": N, "K": K}
        state = {"i": 0}
        ref = None
        for tag, tile in TILES.items():
            if tag == "pp" and (N % 256 or K % 128):
                continue

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
