"""Round 5: which tile geometry for the few-row GEMMs?  Every ring geometry (32x32, 64x64, 64x64 K2, 128x64, 64x128, 128x128) on the
GEMMs of one frame / 2-8 frames, one audio segment, 1-9 questions, COLD weights (a new weight copy per call, more copies than the
Infinity Cache holds), bit equality with the 128x128 double-buffered kernel, and the same for the split-K launches of fc2.
usage: rect_tile_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, event_ms

L, lib = load_probe()
EPI = {"bias": 0, "gelu": 1, "resid": 2}
TILES = {"db128": 0, "auto": -1, "ring32": 8, "ring64": 7, "ring64_k2": 11, "ring128": 6, "r128x64": 12, "r64x128": 13, "r128x128w8": 14, "r128x64w8": 15}
rows = []
st = L.stream_ptr()
SHAPES = [("vision qkv", 3840, 1280, "bias", (257, 514, 1028, 2056)), ("vision fc1", 5120, 1280, "gelu", (257, 514, 1028, 2056)),
          ("vision out", 1280, 1280, "resid", (257, 514, 1028, 2056)), ("vision fc2", 1280, 5120, "resid", (257, 514, 1028, 2056)),
          ("audio qkv", 2304, 768, "bias", (687, 1374)), ("audio fc1", 3072, 768, "gelu", (687, 1374)),
          ("audio out", 768, 768, "resid", (687, 1374)), ("audio fc2", 768, 3072, "resid", (687, 1374)),
          ("text qkv", 3072, 1024, "bias", (77, 308, 693)), ("text fc1", 4096, 1024, "gelu", (77, 308, 693)),
          ("text out", 1024, 1024, "resid", (77, 308, 693)), ("text fc2", 1024, 4096, "resid", (77, 308, 693))]
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
            if tag == "ring32" and ((M + 31) // 32) * (N // 32) > 2048:
                continue
            if tag == "ring64_k2" and ((M + 63) // 64) * (N // 64) > 600:
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
        if name.endswith("fc2"):
            S = 4 if name.startswith("text") else 2
            part = torch.empty(S, M, N, device="cuda")
            pref = None
            for tag, tile in TILES.items():
                if tag in ("db128", "r128x128w8", "r128x64w8") or (tag == "ring32" and ((M + 31) // 32) * (N // 32) * S > 4096) or \
                        (tag == "ring64_k2" and (((M + 63) // 64) * (N // 64) * S > 600 or (K // S) % 128)):
                    continue

                def split():
                    w = ws[state["i"] % copies]
                    state["i"] += 1
                    L.check(lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), M, N, K, S, tile, st), "splitk")
                rec[f"us_split{S}_" + tag] = round(event_ms(split, 3 * copies, warmup=copies) * 1e3, 2)
                state["i"] = 0
                split()
                torch.cuda.synchronize()
                if pref is None:
                    pref = part.clone()
                elif not torch.equal(part, pref):
                    rec[f"DIFFERENT_BITS_split_" + tag] = True
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        if len(sys.argv) > 1:
            json.dump(rows, open(sys.argv[1], "w"), indent=1)
    del ws
