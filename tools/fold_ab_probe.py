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
configs = [("ln_kernel", dict(fold=0, stats=1, skip_tail=0)),
           ("fold_pass", dict(fold=1, stats=0, skip_tail=0)),
           ("fold_epi", dict(fold=1, stats=1, skip_tail=0)),
           ("fold_epi_notail", dict(fold=1, stats=1, skip_tail=1)),
           ("ln_kernel_notail", dict(fold=0, stats=1, skip_tail=1))]
res = {}
for B in (256, 32, 64, 128):
    x = torch.randn(B, 3, 224, 224, device="cuda"); out = torch.empty(B, 1024, device="cuda")
    times = {n: [] for n, _ in configs}
    for rnd in range(4):
        for name, c in configs:
            tower.set_folded_layernorm(bool(c["fold"]))
            setter(lib, "g_enc_fold_stats")(c["stats"])
            setter(lib, "g_gemm_skip_tail")(c["skip_tail"])
            times[name].append(event_ms(lambda: tower.forward_into(x, out), 4 if B >= 128 else 10, warmup=2))
    for name, _ in configs:
        t = sorted(times[name])
        res[f"B{B}_{name}"] = {"ms_median": round((t[1] + t[2]) / 2, 3), "ms_min": round(t[0], 3), "img_per_s": round(B / ((t[1] + t[2]) / 2) * 1e3)}
        print(f"B={B} {name:18s} median {res[f'B{B}_{name}']['ms_median']:8.3f} ms  min {t[0]:8.3f}  {res[f'B{B}_{name}']['img_per_s']} img/s", flush=True)
setter(lib, "g_gemm_skip_tail")(0); setter(lib, "g_enc_fold_stats")(1); tower.set_folded_layernorm(True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
