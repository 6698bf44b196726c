"""Race hunt for the few-row GEMM kernels: every small-launch geometry, many shapes, hundreds of repetitions each, with a large GEMM
running on a second stream to perturb the timing; every result must have the bits of the double-buffered 128x128 kernel.
usage: python tools/ring_stress.py [repetitions]"""
import sys
import torch
from probe_common import load_probe

L, lib = load_probe()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
side = torch.cuda.Stream()
big_a = torch.randn(8192, 1280, device="cuda").to(torch.bfloat16)
big_w = torch.randn(5120, 1280, device="cuda").to(torch.bfloat16)
big_b = torch.zeros(5120, device="cuda")
big_c = torch.empty(8192, 5120, dtype=torch.bfloat16, device="cuda")
bad = 0
shapes = [(77, 1024, 4096), (77, 3072, 1024), (1, 1024, 1024), (16, 1280, 5120), (100, 768, 3072), (257, 1280, 1280), (154, 4096, 1024),
          (308, 1024, 4096), (33, 3840, 1280), (64, 128, 64), (65, 256, 128), (640, 1280, 640)]
for M, N, K in shapes:
    g = torch.Generator(device="cuda").manual_seed(M * 31 + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    x0 = torch.randn(M, N, device="cuda", generator=g)
    for epi, dtype in ((2, torch.float32), (1, torch.bfloat16)):
        ref = x0.clone().to(dtype)
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), ref.data_ptr(), M, N, K, epi, 0, L.stream_ptr()), "ref")
        for tile in (5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, -1):     # incl. round 5's 128x64 / 64x128 and eight-wave ring tiles
            if tile in (9, 11) and (K // 64) % 2:
                continue
            if tile == 10 and (K // 64) % 4:
                continue
            diff = 0
            for r in range(REPS):
                if r % 2 == 0:
                    with torch.cuda.stream(side):
                        L.check(lib.hmm_op_gemm_bf16_tile(big_a.data_ptr(), big_w.data_ptr(), big_b.data_ptr(), big_c.data_ptr(), 8192, 5120, 1280, 1, 3,
                                                          side.cuda_stream), "load")
                c = x0.clone().to(dtype)
                L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, tile, L.stream_ptr()), "gemm")
                diff += int(not torch.equal(c, ref))
            if diff:
                bad += diff
                print(f"M={M} N={N} K={K} epi={epi} tile={tile}: {diff} of {REPS} runs differ", flush=True)
    # round 5: the split-K slabs + the reducing LayerNorm under the same perturbation (bits depend on (K, splits) only)
    for S in (2, 4):
        if K % (128 * S) or N not in (768, 1024, 1280):
            continue
        gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        ref_x = ref_y = None
        for tile in (-1, 6, 7, 8, 11, 12, 13, 14, 15, 16):
            diff = 0
            for r in range(REPS // 2):
                if r % 2 == 0:
                    with torch.cuda.stream(side):
                        L.check(lib.hmm_op_gemm_bf16_tile(big_a.data_ptr(), big_w.data_ptr(), big_b.data_ptr(), big_c.data_ptr(), 8192, 5120, 1280, 1, 3,
                                                          side.cuda_stream), "load")
                part = torch.empty(S, M, N, device="cuda")
                x, y = x0.clone(), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
                L.check(lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), M, N, K, S, tile, L.stream_ptr()), "splitk")
                L.check(lib.hmm_op_layernorm_reduce_bf16(x.data_ptr(), part.data_ptr(), S, bias.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                                         y.data_ptr(), M, N, 1e-6, L.stream_ptr()), "ln_reduce")
                if ref_x is None:
                    ref_x, ref_y = x.clone(), y.clone()
                diff += int(not (torch.equal(x, ref_x) and torch.equal(y, ref_y)))
            if diff:
                bad += diff
                print(f"M={M} N={N} K={K} split-K {S} tile={tile}: {diff} of {REPS // 2} runs differ", flush=True)
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: done", flush=True)
print("ring stress:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
