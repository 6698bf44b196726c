"""Mid-size batches: one chain against two half-batch chains (two streams).  usage: split_min_probe.py"""
import time
from probe_common import load_probe, setter
import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict
set_min0 = setter(lib, "g_enc_split_min")
set_min1 = setter(lib, "g_enc_split_min_audio")


def set_min(v):
    set_min0(v)
    set_min1(v)


def wall_ms(fn, iters=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


for kind, batches in (("vision", (8, 16, 24, 32, 48, 64, 96)), ("audio", (4, 8, 16, 32)), ("text", (16, 32, 64, 128))):
    for B in batches:
        res = {}
        torch.manual_seed(B)
        if kind == 'text':
            x = torch.randint(1, 49000, (B, 77), device='cuda'); x[:, 0], x[:, 20] = 49406, 49407
        else:
            x = torch.randn(B, 3, 224, 224, device='cuda') if kind == 'vision' else torch.randn(B, 3, 1, 128, 204, device='cuda')
        for m in (1 << 30, 2):
            set_min(m)
            tower = HipTower(kind, synthetic_state_dict((kind,), seed=99))     # the workspace is sized at creation / first use
            out = torch.empty(B, 1024, device="cuda")
            res[m] = (wall_ms(lambda: tower.forward_into(x, out)), out.clone())
            del tower
        print(f"{kind} B={B}: one chain {res[1 << 30][0]:.3f} ms   two chains {res[2][0]:.3f} ms   same bits {torch.equal(res[1 << 30][1], res[2][1])}", flush=True)
set_min0(16)
set_min1(12)
