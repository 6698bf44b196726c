"""The split-K launches of the few-row regime (fc2 of one frame, one audio segment, 1-9 questions): the launcher's choice against every
ring geometry, alone with COLD weights, slabs compared bitwise.  usage: splitk_tile_audit_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, event_ms

L, lib = load_probe()
st = L.stream_ptr()
TILES = {"auto": -1, "ring32": 8, "ring32_k2": 9, "ring32_k4": 10, "ring64": 7, "ring64_k2": 11, "ring128": 6, "r128x64": 12, "r64x128": 13,
         "r128x128w8": 14, "r128x64w8": 15, "r64x128w8": 16}
rows = []
for name, N, K, S, Ms in (("vision fc2", 1280, 5120, 2, (257,)), ("audio fc2", 768, 3072, 2, (687,)), ("text fc2", 1024, 4096, 4, (77, 154, 308, 462, 693))):
    copies = max(4, int(400e6 // (N * K * 2)) + 1)
    g = torch.Generator(device="cuda").manual_seed(N + K)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(copies)]
    for M in Ms:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        part = torch.empty(S, M, N, device="cuda")
        rec = {"gemm": name, "M": M, "N": N, "K": K, "splits": S}
        state = {"i": 0}
        ref = None
        for rep in range(2):
            for tag, tile in TILES.items():
                if (tag == "ring32_k2" and (K // S // 64) % 2) or (tag == "ring32_k4" and (K // S // 64) % 4) or (tag == "ring64_k2" and (K // S // 64) % 2):
                    continue

                def call():
                    w = ws[state["i"] % copies]
                    state["i"] += 1
                    L.check(lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), M, N, K, S, tile, st), "splitk")
                rec["us_" + tag] = round(event_ms(call, 3 * copies, warmup=copies) * 1e3, 2)
                state["i"] = 0
                call()
                torch.cuda.synchronize()
                if ref is None:
                    ref = part.clone()
                elif not torch.equal(part, ref):
                    rec["DIFFERENT_BITS_" + tag] = True
        best = min((v, t) for t, v in rec.items() if t.startswith("us_") and t != "us_auto")
        rec["best"], rec["auto_over_best"] = best[1][3:], round(rec["us_auto"] / best[0], 3)
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        if len(sys.argv) > 1:
            json.dump(rows, open(sys.argv[1], "w"), indent=1)
    del ws
