"""The block's GEMMs at the row counts of a 16- and a 32-frame forward, per tile geometry (alone in a loop).  usage: mid_gemm_probe.py"""
import json
import sys

import torch

from probe_common import load_probe, event_ms

L, lib = load_probe()
TILES = {"128x128": 0, "256x128": 1, "256x256": 2, "pp": 3, "ring128": 6, "auto_tiled": -2}
for name, N, K, epi in (("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2), ("qkv", 3840, 1280, 0)):
    for M in (4112, 8224, 16448):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        bias = torch.zeros(N, device="cuda")
        c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
        rec = {"gemm": name, "M": M, "N": N, "K": K}
        for tag, tile in TILES.items():
            ms = event_ms(lambda: L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi,
                                                                    tile, L.stream_ptr()), "gemm"), 30, warmup=5)
            rec["us_" + tag] = round(ms * 1e3, 1)
            rec["tf_" + tag] = round(2.0 * M * N * K / ms / 1e9, 0)
        print(json.dumps(rec), flush=True)
