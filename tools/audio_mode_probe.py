"""Audio tower at 4 ... 16 segments: {one, two chains} x {fused in_proj + attention, projection GEMM + attention kernel}, interleaved three times,
bit equality against the shipped setting.  usage: audio_mode_probe.py [out.json] [sizes,comma]"""
import json
import sys
import time

import torch

from probe_common import load_probe, setter

L, lib = load_probe()
out_path = sys.argv[1] if len(sys.argv) > 1 else None
sizes = tuple(int(b) for b in sys.argv[2].split(",")) if len(sys.argv) > 2 else (4, 5, 6, 7, 8, 9, 10, 12, 16)
split_min, fused_min, one_round = setter(lib, "g_enc_split_min_audio"), setter(lib, "g_enc_fused_min_audio"), setter(lib, "g_enc_audio_one_round")
from hippomm_amd.encoder import HipTower, synthetic_state_dict   # noqa: E402

CONFIGS = {"shipped": (12, 9, 1), "one_chain_fused": (9999, 9, 1), "two_chains_fused": (12, 9, 0), "one_chain_unfused": (9999, 9999, 1),
           "two_chains_unfused": (12, 9999, 0)}


def apply(cfg):
    split_min(cfg[0]); fused_min(cfg[1]); one_round(cfg[2])


def wall_ms(fn, iters):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


tower = HipTower("audio", synthetic_state_dict(("audio",), seed=99))
rows = []
for B in sizes:
    x = torch.randn(B, 3, 1, 128, 204, device="cuda")
    out = torch.empty(B, 1024, device="cuda")
    ms = {k: [] for k in CONFIGS}
    outs = {}
    for rep in range(3):
        for k, cfg in CONFIGS.items():
            apply(cfg)
            ms[k].append(wall_ms(lambda: tower.forward_into(x, out), 30))
            outs[k] = out.clone()
    rec = {"segments": B}
    for k in CONFIGS:
        rec["ms_" + k] = round(min(ms[k]), 4)
        if not torch.equal(outs[k], outs["shipped"]):
            rec["DIFFERENT_BITS_" + k] = True
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    if out_path:
        json.dump(rows, open(out_path, "w"), indent=1)
apply(CONFIGS["shipped"])
