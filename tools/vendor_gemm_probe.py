"""Reference point only (never on the product path): the vendor BLAS behind torch.matmul on the four ViT-H GEMM shapes,
bf16 inputs, fp32 accumulate, bf16 output, no bias / activation / residual epilogue -- next to gemm_bf16 with its fused
epilogues on the same shapes."""
import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from hippomm_amd import _lib as L
lib = L.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65792


def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for name, N, K, epi in [("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    t_vendor = timeit(lambda: torch.matmul(a, w.t(), out=out))
    t_ours = timeit(lambda: L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi,
                                                         L.stream_ptr()), "gemm"))
    fl = 2.0 * M * N * K
    print(f"{name} M={M} N={N} K={K}: vendor matmul (no epilogue) {t_vendor:.3f} ms = {fl/t_vendor/1e9:.0f} TFLOP/s | "
          f"gemm_bf16 (epilogue {epi}) {t_ours:.3f} ms = {fl/t_ours/1e9:.0f} TFLOP/s", flush=True)
