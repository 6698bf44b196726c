"""Wall clock per forward of one tower over a list of batch sizes on the PRODUCT library (median of 7 x 20 calls after 20 warm-up
calls): the per-size record the dispatch rules are judged by.  usage: forward_sweep_probe.py <vision|audio|text> <sizes,comma> [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                       # noqa: E402
from hippomm_amd.encoder import HipTower, synthetic_state_dict    # noqa: E402

kind, sizes = sys.argv[1], [int(b) for b in sys.argv[2].split(",")]
tower = HipTower(kind, synthetic_state_dict((kind,), seed=1234))
gen = torch.Generator(device="cuda").manual_seed(5)
rows = []
for b in sizes:
    if kind == "vision":
        x = torch.randn(b, 3, 224, 224, device="cuda", generator=gen)
    elif kind == "audio":
        x = torch.randn(b, 3, 1, 128, 204, device="cuda", generator=gen)
    else:
        x = torch.randint(1, 49000, (b, 77), device="cuda", generator=gen)
        x[:, 0], x[:, 20] = 49406, 49407
    emb = torch.empty(b, 1024, device="cuda")
    for _ in range(20):
        tower.forward_into(x, emb)
    torch.cuda.synchronize()
    t = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(20):
            tower.forward_into(x, emb)
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / 20 * 1e3)
    rec = {"tower": kind, "batch": b, "ms": round(sorted(t)[3], 4), "per_sample_ms": round(sorted(t)[3] / b, 4)}
    rows.append(rec)
    print(json.dumps(rec), flush=True)
if len(sys.argv) > 3:
    json.dump(rows, open(sys.argv[3], "w"), indent=1)
