"""Where does a K-step of the ring kernels go?  Timing-only ablations (probe build; results are wrong): the eight-wave 128 x 128
ring and the 64 x 64 ring without MFMAs / without fragment reads and MFMAs / without the LDS-DMA, and 3 / 5 ring stages, on
one-frame and ten-frame GEMMs with COLD and with WARM (one copy, L2 / Infinity-Cache resident) weights.
usage: ring_ablation_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, event_ms

L, lib = load_probe()
st = L.stream_ptr()
T128 = {"full": 14, "no_mfma": 101, "no_reads_no_mfma": 102, "no_dma": 103, "5_stages": 105, "3_stages": 106}
T64 = {"full": 7, "no_mfma": 111, "no_reads_no_mfma": 112, "no_dma": 113}
rows = []
for name, N, K, Ms in (("vision fc2", 1280, 5120, (257, 2570)), ("vision fc1", 5120, 1280, (257, 2570)), ("vision qkv", 3840, 1280, (257,))):
    for cold in (True, False):
        copies = max(4, int(400e6 // (N * K * 2)) + 1) if cold else 1
        g = torch.Generator(device="cuda").manual_seed(N + K)
        ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(copies)]
        bias = torch.randn(N, device="cuda", generator=g)
        for M in Ms:
            a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            for fam, tiles in (("128x128w8", T128), ("64x64", T64)):
                rec = {"gemm": name, "M": M, "N": N, "K": K, "weights": "cold" if cold else "warm", "tile": fam, "k_tiles": K // 64}
                state = {"i": 0}
                for tag, tile in tiles.items():
                    def call():
                        w = ws[state["i"] % copies]
                        state["i"] += 1
                        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, 0, tile, st), "gemm")
                    try:
                        rec["us_" + tag] = round(event_ms(call, max(3 * copies, 60), warmup=max(copies, 20)) * 1e3, 2)
                    except Exception as e:                          # e.g. five stages = all 160 KiB of LDS
                        rec["us_" + tag] = str(e)[:80]
                rows.append(rec)
                print(json.dumps(rec), flush=True)
                if len(sys.argv) > 1:
                    json.dump(rows, open(sys.argv[1], "w"), indent=1)
        del ws
