"""Round 5: the staggered 256 x 128 ring tile in the vision forwards of 12-48 frames: every combination of {two chains from 13 / 17 /
21 frames} x {ring256 off / for launches of < 100 ping-pong tiles} x {only past 256 ring tiles / past 128}, interleaved three times,
bit equality against the shipped setting.  usage: ring256_forward_probe.py [out.json] [sizes,comma]"""
import json
import sys
import time

import torch

from probe_common import load_probe, setter

L, lib = load_probe()
out_path = sys.argv[1] if len(sys.argv) > 1 else None
sizes = tuple(int(b) for b in sys.argv[2].split(",")) if len(sys.argv) > 2 else (12, 13, 14, 15, 16, 18, 20, 24, 26, 28, 32, 36, 40, 48)
split_min, pp_tiles, min_r128 = setter(lib, "g_enc_split_min"), setter(lib, "g_gemm_ring256_pp_tiles"), setter(lib, "g_gemm_ring256_min_r128")
from hippomm_amd.encoder import HipTower, synthetic_state_dict   # noqa: E402

CONFIGS = {"shipped": (13, 0, 256), "ring256": (13, 100, 256), "ring256_r128": (13, 100, 128),
           "one_chain_to_16": (17, 0, 256), "one_chain_to_16+ring256": (17, 100, 256),
           "one_chain_to_20+ring256": (21, 100, 256)}


def apply(cfg):
    split_min(cfg[0]); pp_tiles(cfg[1]); min_r128(cfg[2])


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
for B in sizes:
    x = torch.randn(B, 3, 224, 224, device="cuda")
    out = torch.empty(B, 1024, device="cuda")
    ms = {k: [] for k in CONFIGS}
    outs = {}
    for rep in range(3):
        for k, cfg in CONFIGS.items():
            apply(cfg)
            ms[k].append(wall_ms(lambda: tower.forward_into(x, out), 20))
            outs[k] = out.clone()
    rec = {"frames": B}
    for k in CONFIGS:
        rec["ms_" + k] = round(min(ms[k]), 4)
        if not torch.equal(outs[k], outs["shipped"]):
            rec["DIFFERENT_BITS_" + k] = True
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    if out_path:
        json.dump(rows, open(out_path, "w"), indent=1)
apply(CONFIGS["shipped"])
