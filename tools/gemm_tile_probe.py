"""Which tile geometry of gemm_bf16 is fastest at the row counts of small batches?  (hmm_op_gemm_bf16_tile: 0 = 128x128,
1 = 256x128, 2 = 256x256 single-phase, 3 = 256x256 ping-pong, 4 = ping-pong with the peeled tail = the default.)
usage: gemm_tile_probe.py [images ...]   (rows = images * 257)"""
import sys
from probe_common import load_probe, event_ms
import torch

L, lib = load_probe()
images = [int(v) for v in sys.argv[1:]] or [8, 16, 32, 48, 64]
shapes = [("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2), ("qkv", 3840, 1280, 0)]
for B in images:
    M = B * 257
    for name, N, K, epi in shapes:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        c = torch.randn(M, N, device="cuda").to(torch.float32 if epi == 2 else torch.bfloat16)
        row = []
        for tile in (0, 1, 2, 3, 4):
            ms = event_ms(lambda: L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(),
                                                                     M, N, K, epi, tile, L.stream_ptr()), "gemm"), 20, warmup=3)
            row.append(f"{ms*1e3:6.1f}")
        print(f"B={B:3d} M={M:6d} {name:4s} us by tile [128x128, 256x128, 256x256, pp, pp+peel]: {' '.join(row)}", flush=True)
