"""Uneven two-chain split of the 256-frame batch (tile-round quantisation of the N = 1280 GEMMs): forward time per split,
interleaved rounds, plus a bitwise check against the even split."""
from probe_common import load_probe, setter, event_ms
import torch
L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
torch.cuda.empty_cache()
x = torch.randn(256, 3, 224, 224, device="cuda"); out = torch.empty(256, 1024, device="cuda")
splits = [128, 102, 101, 153, 154, 112, 96, 64]
tower.forward_into(x, out); ref = out.clone()
times = {s: [] for s in splits}
for rnd in range(4):
    for sp in splits:
        setter(lib, "g_enc_split_num")(sp)
        times[sp].append(event_ms(lambda: tower.forward_into(x, out), 4, warmup=2))
        if rnd == 0:
            torch.cuda.synchronize()
            print(f"split {sp}: bitwise equal to 128/128: {torch.equal(out, ref)}", flush=True)
for sp in splits:
    t = sorted(times[sp])
    print(f"split {sp:3d}/{256 - sp:3d}: median {(t[1] + t[2]) / 2:.3f} ms  min {t[0]:.3f}  {256 / ((t[1] + t[2]) / 2) * 1e3:.0f} img/s", flush=True)
setter(lib, "g_enc_split_num")(128)
