"""Few-row GEMMs (one question through the text tower, cls rows, single frames): the tiled kernels against the one-wave sliver
kernel (HMM_GEMM_TILE_SLIVER) at 16 / 32 / 64 rows per wave, per shape, and bit equality of the two.  Then the text tower at
batch 1 .. 4 end to end, eager and as a replayed graph.  Usage: python tools/sliver_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import load_probe, setter, event_ms

L, lib = load_probe()
set_mt = setter(lib, "g_gemm_sliver_mt")
EPI = {"bias": 0, "gelu": 1, "resid": 2}
rows = []


def run(M, N, K, epi, tile):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(M, N, device="cuda", generator=g)
    c = c0.clone() if epi == "resid" else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    st = L.stream_ptr()

    def call():
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, EPI[epi], tile, st), "gemm")
    ms = event_ms(call, 200, warmup=20)
    if epi == "resid":
        c.copy_(c0)
    call()
    torch.cuda.synchronize()
    return ms * 1e3, c.clone()


SHAPES = [("text qkv", 3072, 1024, "bias"), ("text out", 1024, 1024, "resid"), ("text fc1", 4096, 1024, "gelu"),
          ("text fc2", 1024, 4096, "resid"), ("vision qkv", 3840, 1280, "bias"), ("vision fc1", 5120, 1280, "gelu"),
          ("vision fc2", 1280, 5120, "resid"), ("audio fc1", 3072, 768, "gelu"), ("audio fc2", 768, 3072, "resid")]
for name, N, K, epi in SHAPES:
    for M in ((77, 154, 308, 1232) if name.startswith("text") else (16, 128, 192, 257, 384)):
        us0, c0 = run(M, N, K, epi, 0)
        usa, ca = run(M, N, K, epi, -1)
        usr, cr = run(M, N, K, epi, 6)
        us64, c64 = run(M, N, K, epi, 7)
        us32, c32 = run(M, N, K, epi, 8)
        rec = {"gemm": name, "M": M, "N": N, "K": K, "us_128x128": round(us0, 1), "us_128x128_ring4": round(usr, 1),
               "ring_same_bits": bool(torch.equal(cr, c0)), "us_64x64_ring4": round(us64, 1),
               "ring64_same_bits": bool(torch.equal(c64, c0)), "us_32x32_ring4": round(us32, 1),
               "ring32_same_bits": bool(torch.equal(c32, c0)), "us_auto": round(usa, 1)}
        for mt in (1, 2, 4):
            set_mt(mt)
            us, c = run(M, N, K, epi, 5)
            rec[f"us_sliver_{16 * mt}"] = round(us, 1)
            rec[f"same_bits_{16 * mt}"] = bool(torch.equal(c, c0))
        set_mt(0)
        rows.append(rec)
        print(json.dumps(rec), flush=True)

if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
