"""A/B the ping-pong GEMM experiment flags (EPI_BIAS_BF16 only), interleaved rounds in one process."""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from hippomm_amd import _lib as L
lib = L.load()
lib.hmm_dev_gemm_bf16_variant.restype = C.c_int
lib.hmm_dev_gemm_bf16_variant.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
M = 65792
variants = [int(v) for v in sys.argv[1].split(",")]
shapes = [("N3840_K1280", 3840, 1280), ("N1280_K5120", 1280, 5120)]
for name, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    res = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            def run():
                L.check(lib.hmm_dev_gemm_bf16_variant(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(),
                                                      M, N, K, 0, v, L.stream_ptr()), "gemm")
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 8)
    for v in variants:
        ms = sorted(res[v])
        print(f"{name} variant={v:4d}: median {ms[len(ms)//2]:.3f} ms min {ms[0]:.3f}  {2*M*N*K/ms[len(ms)//2]/1e9:.0f} TF (median)", flush=True)
