#!/usr/bin/env python3
"""Do two forwards of the vision tower overlap when they are issued on two streams (two handles)?  The files -> embeddings
pipeline cuts a call into ranges; back to back on one stream each range pays the latency-bound start of a small forward
(f(n) ~ 3.2 ms + 0.29 ms per frame).  Here: the same ranges alternating between two handles on two streams.

    python tools/concurrent_forward_probe.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from hippomm_amd.encoder import HipTower, synthetic_state_dict


def main():
    sd = synthetic_state_dict(("vision",), seed=1234)
    towers = [HipTower("vision", sd), HipTower("vision", sd)]
    del sd
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    x = torch.randn(256, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    emb = torch.empty(256, 1024, device="cuda")
    towers[0].forward_into(x, emb)
    want = emb.clone()
    rows = []

    def run(plan, two):
        cur = torch.cuda.current_stream()
        lo = 0
        for s in streams:
            s.wait_stream(cur)
        for k, c in enumerate(plan):
            t = towers[k % 2] if two else towers[0]
            with torch.cuda.stream(streams[k % 2] if two else streams[0]):
                t.forward_into(x[lo:lo + c], emb[lo:lo + c])
            lo += c
        for s in streams:
            cur.wait_stream(s)

    for plan in ([256], [128, 128], [64] * 4, [32] * 8, [16] * 16, [14, 28, 56, 112, 46], [16, 16, 25, 36, 49, 63, 51], [32, 32], [16, 16], [14, 18]):
        for two in (False, True):
            if len(plan) == 1 and two:
                continue
            for _ in range(2):
                run(plan, two)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                emb.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(plan, two)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            n = sum(plan)
            rows.append({"plan": plan, "two_streams": two, "ms": round(sorted(ts)[2], 3), "same_bits": bool(torch.equal(emb[:n], want[:n]))})
            print(json.dumps(rows[-1]), flush=True)
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
