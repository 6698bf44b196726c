"""fc1 / qkv GEMM at M=65536 under two tile walks, a few launches each (for rocprofv3 --pmc FETCH_SIZE) + timing."""
import sys
from probe_common import load_probe, setter, event_ms
import torch
L, lib = load_probe()
M = 65536
for name, N, K, epi in (("fc1", 5120, 1280, 1), ("qkv", 3840, 1280, 0)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda: L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, 3, L.stream_ptr()), "g")
    for walk in (0, 2052, 1032, 8200, 264):          # strips, 8x4, 4x8, 32x8, 1x8
        setter(lib, "g_gemm_walk")(walk)
        ms = event_ms(run, 10, warmup=3)
        print(f"{name} walk={walk >> 8}x{walk & 255}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.0f} TF", flush=True)
