"""Two chains of UNEVEN size for 13 ... 24 frames: the first chain takes b0 frames (12 = the largest size that still runs well as one chain,
B - 6, B / 2 = shipped), the second the rest.  Interleaved, bit equality.  usage: uneven_chains_probe.py [out.json]"""
import json
import sys
import time

import torch

from probe_common import load_probe, setter

L, lib = load_probe()
out_path = sys.argv[1] if len(sys.argv) > 1 else None
split_num = setter(lib, "g_enc_split_num")
from hippomm_amd.encoder import HipTower, synthetic_state_dict   # noqa: E402


def wall_ms(fn, iters):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


tower = HipTower("vision", synthetic_state_dict(("vision",), seed=99))
rows = []
for B in (13, 14, 15, 16, 18, 20, 24):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    out = torch.empty(B, 1024, device="cuda")
    cands = {}
    for b0 in sorted({B // 2, 12, B - 6, B - 4, B - 1, 8}):
        if 1 <= b0 < B:
            num = -(-256 * b0 // B)                       # ceil: (B * num) // 256 == b0
            while (B * num) // 256 > b0:
                num -= 1
            if (B * num) // 256 == b0:
                cands[b0] = num
    ms = {b0: [] for b0 in cands}
    outs = {}
    for rep in range(3):
        for b0, num in cands.items():
            split_num(num)
            ms[b0].append(wall_ms(lambda: tower.forward_into(x, out), 20))
            outs[b0] = out.clone()
    rec = {"frames": B}
    for b0 in cands:
        rec[f"ms_first_chain_{b0}"] = round(min(ms[b0]), 4)
        if not torch.equal(outs[b0], outs[B // 2]):
            rec[f"DIFFERENT_BITS_{b0}"] = True
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    if out_path:
        json.dump(rows, open(out_path, "w"), indent=1)
split_num(128)
