"""A/B in ONE process, interleaved rounds, probe build: the ViT-H forward (two streams) at 256 and 32 frames with
  fold=0            LayerNorm as its own kernel
  fold=1 stats=0    LayerNorm folded, row statistics by a pass over xb (round 2's branch)
  fold=1 stats=1    LayerNorm folded, row statistics from the residual epilogues (chunk sums + finalize)
  ... skip_tail=1   the same without the peeled GEMM tails (timing upper bound of folding the tails into the main launches)
usage: fold_ab_probe.py [json_out]"""
import json
import sys
from probe_common import load_probe, setter, event_ms
import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict

tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
torch.cuda.empty_cache()
KNOBS = ("g_enc_fold_stats", "g_gemm_skip_tail", "g_ln_nt_loads", "g_gemm_walk")
BASE = dict(fold=0, g_enc_fold_stats=1, g_gemm_skip_tail=0, g_ln_nt_loads=1, g_gemm_walk=-1)      # the product's defaults
W84, W65 = (8 << 8) | 4, (6 << 8) | 5
configs_big = [("product", {}), ("ln_plain_loads", dict(g_ln_nt_loads=0)), ("walk_strips", dict(g_gemm_walk=0)),
               ("walk_8x4_all", dict(g_gemm_walk=W84)), ("walk_6x5_all", dict(g_gemm_walk=W65)),
               ("round2_like", dict(g_ln_nt_loads=0, g_gemm_walk=0)),
               ("fold_epi", dict(fold=1)), ("fold_pass", dict(fold=1, g_enc_fold_stats=0))]
configs_small = [("product", {}), ("fold_epi", dict(fold=1))]
res = {}
for B in (256, 32, 128):
    configs = configs_big if B == 256 else configs_small
    x = torch.randn(B, 3, 224, 224, device="cuda"); out = torch.empty(B, 1024, device="cuda")
    times = {n: [] for n, _ in configs}
    for rnd in range(4):
        for name, c in configs:
            c = dict(BASE, **c)
            tower.set_folded_layernorm(bool(c["fold"]))
            for k in KNOBS:
                setter(lib, k)(c[k])
            times[name].append(event_ms(lambda: tower.forward_into(x, out), 4 if B >= 128 else 10, warmup=2))
    for name, _ in configs:
        t = sorted(times[name])
        res[f"B{B}_{name}"] = {"ms_median": round((t[1] + t[2]) / 2, 3), "ms_min": round(t[0], 3), "img_per_s": round(B / ((t[1] + t[2]) / 2) * 1e3)}
        print(f"B={B} {name:20s} median {res[f'B{B}_{name}']['ms_median']:8.3f} ms  min {t[0]:8.3f}  {res[f'B{B}_{name}']['img_per_s']} img/s", flush=True)
for k in KNOBS:
    setter(lib, k)(BASE[k])
tower.set_folded_layernorm(False)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
