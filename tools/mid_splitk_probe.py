"""Round 5: the MID regime -- vision forwards of 8 ... 64 frames (the reference's segment and 32-frame-buffer calls) with fc2 as
split-K on the ping-pong tile, reduced by the next norm_1 (out of place in the fused path), against the unsplit forward.
Wall clock per forward, interleaved twice, cosine against the unsplit embeddings.  usage: mid_splitk_probe.py [out.json]"""
import json
import sys
import time

import torch

from probe_common import load_probe, setter

L, lib = load_probe()
set_hi = setter(lib, "g_enc_splitk_mid_hi")
set_lo = setter(lib, "g_enc_splitk_mid_lo")
set_s = setter(lib, "g_enc_splitk_mid")
from hippomm_amd.encoder import HipTower, synthetic_state_dict   # noqa: E402


def wall_ms(fn, iters=20):
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
for B in (4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 96):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    rec = {"frames": B}
    base = None
    for rep in range(2):
        for tag, lo, hi, S in (("unsplit", 0, 0, 3), ("mid_s3", 257, 1 << 30, 3), ("mid_s2", 257, 1 << 30, 2), ("mid_s4", 257, 1 << 30, 4)):
            set_lo(lo); set_hi(hi); set_s(S)
            out = torch.empty(B, 1024, device="cuda")
            ms = wall_ms(lambda: tower.forward_into(x, out))
            rec[f"ms_{tag}_{rep}"] = round(ms, 3)
            if base is None:
                base = out.clone()
            elif rep == 0:
                o, b = out.double(), base.double()
                rec[f"one_minus_cos_{tag}"] = float((1 - (o * b).sum(1) / (o.norm(dim=1) * b.norm(dim=1))).max())
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)
