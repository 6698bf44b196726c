"""Does replaying a small forward as a HIP graph beat launching it?  Stream launches of dependent kernels cost ~2.6 us each on
MI355X, graph nodes ~1.6 us (tools/grid_barrier_probe.py); a one-question forward is 174 launches.
usage: graph_latency_probe.py [json_out]"""
import json
import sys
import time

from probe_common import load_probe

import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict  # noqa: E402

res = []
for kind, batches in (("text", (1, 2, 4)), ("vision", (1, 2, 4, 8, 32)), ("audio", (1, 2, 4))):
    tower = HipTower(kind, synthetic_state_dict((kind,), seed=1))
    for B in batches:
        if kind == "text":
            x = torch.randint(1, 49000, (B, 77), device="cuda"); x[:, 0], x[:, 20] = 49406, 49407
        else:
            x = torch.randn(B, 3, 224, 224, device="cuda") if kind == "vision" else torch.randn(B, 3, 1, 128, 204, device="cuda")
        out_e = torch.empty(B, 1024, device="cuda")
        out_g = torch.empty(B, 1024, device="cuda")

        def wall(fn, reps=5, inner=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t = []
            for _ in range(reps):
                t0 = time.perf_counter()
                for _ in range(inner):
                    fn()
                torch.cuda.synchronize()
                t.append((time.perf_counter() - t0) / inner * 1e3)
            return sorted(t)[len(t) // 2]

        ms_eager = wall(lambda: tower.forward_into(x, out_e))
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            tower.forward_into(x, out_g)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                tower.forward_into(x, out_g)
        ms_graph = wall(lambda: g.replay())
        torch.cuda.synchronize()
        rec = {"tower": kind, "batch": B, "ms_eager": round(ms_eager, 4), "ms_graph_replay": round(ms_graph, 4),
               "same_bits": bool(torch.equal(out_e, out_g))}
        res.append(rec)
        print(rec, flush=True)
    del tower
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
