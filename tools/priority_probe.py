"""Does a lower / higher HIP priority of the second chain's stream change the two-stream forward?  One tower per setting
(the priority is read when the handle is created), interleaved rounds."""
from probe_common import load_probe, setter, event_ms
import torch
L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict
sd = synthetic_state_dict(("vision",), seed=1234)
towers = {}
for prio in (0, 1, -1):
    setter(lib, "g_enc_side_priority")(prio)
    towers[prio] = HipTower("vision", sd)
setter(lib, "g_enc_side_priority")(0)
del sd
torch.cuda.empty_cache()
x = torch.randn(256, 3, 224, 224, device="cuda"); out = torch.empty(256, 1024, device="cuda")
ref = None
times = {p: [] for p in towers}
for rnd in range(4):
    for p, t in towers.items():
        times[p].append(event_ms(lambda: t.forward_into(x, out), 4, warmup=2))
        if rnd == 0:
            torch.cuda.synchronize()
            ref = out.clone() if ref is None else ref
            print(f"priority {p}: bitwise equal: {torch.equal(out, ref)}", flush=True)
for p in towers:
    t = sorted(times[p])
    print(f"side-stream priority {p:2d}: median {(t[1] + t[2]) / 2:.3f} ms  min {t[0]:.3f}", flush=True)
