"""BASELINE cfg 3 (128 frame + audio pairs) with the audio tower's fused in_proj + attention kernel on / off, and the audio
tower alone; interleaved rounds in one process.  usage: joint_probe.py"""
from probe_common import ROOT, event_ms  # noqa: F401
import torch
from hippomm_amd.encoder import HipTower, synthetic_state_dict

pairs = 128
sd = synthetic_state_dict(("vision", "audio"), seed=1234)
vis, aud = HipTower("vision", sd), HipTower("audio", sd)
del sd
frames = torch.randn(pairs, 3, 224, 224, device="cuda")
mels = torch.randn(pairs, 3, 1, 128, 204, device="cuda")
ev, ea = torch.empty(pairs, 1024, device="cuda"), torch.empty(pairs, 1024, device="cuda")
def both():
    vis.forward_into(frames, ev); aud.forward_into(mels, ea)
res = {}
for rnd in range(3):
    for fused in (1, 0):
        aud.set_fused_attention(bool(fused))
        res.setdefault(("joint", fused), []).append(event_ms(both, 5, warmup=2))
        res.setdefault(("audio", fused), []).append(event_ms(lambda: aud.forward_into(mels, ea), 8, warmup=2))
for (what, fused), t in sorted(res.items()):
    t = sorted(t)
    print(f"{what} fused={fused}: median {t[1]:.3f} ms  min {t[0]:.3f}  -> {pairs / t[1] * 1e3:.0f} pairs/s", flush=True)
