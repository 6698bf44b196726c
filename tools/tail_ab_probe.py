"""A/B in one process: the peeled last row tile of the big GEMM launches on 64x64 tiles vs 128x128 tiles (both behind the ring),
vision at 256 / 128 frames, audio at 128 segments.  usage: tail_ab_probe.py"""
from probe_common import load_probe, setter, event_ms
import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict

set_tail = setter(lib, "g_gemm_tail_64")
for kind, batches in (("vision", (256, 128)), ("audio", (128,))):
    tower = HipTower(kind, synthetic_state_dict((kind,), seed=1234))
    torch.cuda.empty_cache()
    for B in batches:
        x = torch.randn(B, 3, 224, 224, device="cuda") if kind == "vision" else torch.randn(B, 3, 1, 128, 204, device="cuda")
        out = torch.empty(B, 1024, device="cuda")
        times = {0: [], 128: [], 512: []}
        for rnd in range(9):
            for v in (0, 128, 512):
                set_tail(v)
                times[v].append(event_ms(lambda: tower.forward_into(x, out), 5, warmup=2))
        for v in (0, 128, 512):
            t = sorted(times[v])
            print(f"{kind} B={B} tail_64={v}: median {t[4]:8.3f} ms  min {t[0]:8.3f}  max {t[-1]:8.3f}", flush=True)
    del tower
set_tail(128)
