"""Wall clock per forward at the reference's call sizes on the PRODUCT library (one JSON line), for A/B of process-level settings
(HIP runtime environment variables): run it once per setting, see tools/runs/r5env.sh.  usage: call_sizes_probe.py <label>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                       # noqa: E402
from hippomm_amd.encoder import HipTower, synthetic_state_dict    # noqa: E402

rec = {"label": sys.argv[1] if len(sys.argv) > 1 else "",
       "env": {k: v for k, v in os.environ.items() if k.startswith(("HIP_", "ROC_", "AMD_", "DEBUG_", "GPU_", "HSA_"))}}
gen = torch.Generator(device="cuda").manual_seed(5)
for kind, sizes in (("text", (1, 4)), ("audio", (1,)), ("vision", (1, 8, 32))):
    tower = HipTower(kind, synthetic_state_dict((kind,), seed=1234))
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
        rec[f"{kind}_{b}_ms"] = round(sorted(t)[3], 4)
        rec[f"{kind}_{b}_sum"] = float(emb.double().abs().sum())
    del tower
    torch.cuda.empty_cache()
print(json.dumps(rec), flush=True)
