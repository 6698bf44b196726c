"""A/B in ONE process, interleaved rounds, probe build: what the few-row GEMM kernels (4-deep ring for launches of few 128x128
tiles incl. the peeled tails; sliver kernel for few rows) do to the LARGE-batch forwards -- vision at 256 / 128 frames, audio at
128 segments -- where they only carry the peeled last row tile, the cls rows and the head.  usage: small_gemm_ab_probe.py [json]"""
import json
import sys
from probe_common import load_probe, setter, event_ms
import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict

KNOBS = ("g_gemm_small_stages", "g_enc_sliver_rows", "g_gemm_small_64", "g_gemm_small_32")
configs = [("double_buffer_only", dict(g_gemm_small_stages=2, g_enc_sliver_rows=0, g_gemm_small_64=0, g_gemm_small_32=0)),
           ("ring128", dict(g_gemm_small_stages=4, g_enc_sliver_rows=0, g_gemm_small_64=0, g_gemm_small_32=0)),
           ("ring64", dict(g_gemm_small_stages=4, g_enc_sliver_rows=0, g_gemm_small_64=512, g_gemm_small_32=0)),
           ("ring64_sliver_gated", dict(g_gemm_small_stages=4, g_enc_sliver_rows=16448, g_gemm_small_64=512, g_gemm_small_32=0)),
           ("product", dict(g_gemm_small_stages=4, g_enc_sliver_rows=16448, g_gemm_small_64=512, g_gemm_small_32=400))]
res = {}
for kind, batches in (("vision", (256, 128, 64, 32, 16)), ("audio", (128, 16)), ("text", (256, 32))):
    tower = HipTower(kind, synthetic_state_dict((kind,), seed=1234))
    torch.cuda.empty_cache()
    for B in batches:
        if kind == "text":
            x = torch.randint(1, 49000, (B, 77), device="cuda")
            x[:, 0], x[:, 20] = 49406, 49407
        else:
            x = torch.randn(B, 3, 224, 224, device="cuda") if kind == "vision" else torch.randn(B, 3, 1, 128, 204, device="cuda")
        out = torch.empty(B, 1024, device="cuda")
        times = {n: [] for n, _ in configs}
        outs = {}
        for rnd in range(5):
            for name, c in configs:
                for k in KNOBS:
                    setter(lib, k)(c[k])
                times[name].append(event_ms(lambda: tower.forward_into(x, out), 5, warmup=2))
                outs[name] = out.clone()
        for name, _ in configs:
            t = sorted(times[name])
            res[f"{kind}_B{B}_{name}"] = {"ms_median": round(t[2], 3), "ms_min": round(t[0], 3),
                                          "same_bits_as_double_buffer": bool(torch.equal(outs[name], outs["double_buffer_only"]))}
            print(f"{kind} B={B} {name:20s} median {t[2]:8.3f} ms  min {t[0]:8.3f}  same bits {res[f'{kind}_B{B}_{name}']['same_bits_as_double_buffer']}", flush=True)
    del tower
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
