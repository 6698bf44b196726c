"""Small batches: fused in_proj + attention (one 256x256x1280 tile walk per (sample, head): 16 workgroups per frame) against the
few-row projection GEMM + the attention kernel.  Vision 1 .. 32 frames, audio 1 .. 8 segments.  usage: fused_small_probe.py"""
import time
from probe_common import load_probe
import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict


def wall_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


for kind, batches in (("vision", (1, 2, 4, 8, 16, 32, 64)), ("audio", (1, 2, 4, 8, 16))):
    tower = HipTower(kind, synthetic_state_dict((kind,), seed=99))
    for B in batches:
        x = torch.randn(B, 3, 224, 224, device="cuda") if kind == "vision" else torch.randn(B, 3, 1, 128, 204, device="cuda")
        out = torch.empty(B, 1024, device="cuda")
        res, outs = {}, []
        for fused in (1, 0):
            tower.set_fused_attention(bool(fused))
            res[fused] = wall_ms(lambda: tower.forward_into(x, out))
            outs.append(out.clone())
        print(f"{kind} B={B}: fused {res[1]:.3f} ms   separate {res[0]:.3f} ms   same bits {torch.equal(outs[0], outs[1])}", flush=True)
    tower.set_fused_attention(True)
    del tower
