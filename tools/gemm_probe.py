"""Time the bf16 GEMM tile variants on the four ViT-H shapes at B=256 (M=65792)."""
import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from hippomm_amd import _lib as L
lib = L.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65792
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
shapes = [("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2)]
for name, N, K, epi in shapes:
    a = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    for v in variants:
        def run():
            L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(),
                                                  M, N, K, epi, v, L.stream_ptr()), "gemm")
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        print(f"{name:4s} M={M} N={N} K={K} variant={v}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
