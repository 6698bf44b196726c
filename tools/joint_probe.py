"""BASELINE cfg 3 (128 frame + audio pairs): vision then audio on one stream (as round 2 timed it) against the two towers
issued on two streams (each still forks its own second chain), interleaved rounds in one process.  usage: joint_probe.py"""
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
side = torch.cuda.Stream()
vis.forward_into(frames, ev); aud.forward_into(mels, ea); torch.cuda.synchronize()
ref_v, ref_a = ev.clone(), ea.clone()


def serial():
    vis.forward_into(frames, ev); aud.forward_into(mels, ea)


def concurrent():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        aud.forward_into(mels, ea)
    vis.forward_into(frames, ev)
    main.wait_stream(side)


res = {"serial": [], "concurrent": []}
for rnd in range(4):
    for name, fn in (("serial", serial), ("concurrent", concurrent)):
        res[name].append(event_ms(fn, 5, warmup=2))
        torch.cuda.synchronize()
        assert torch.equal(ev, ref_v) and torch.equal(ea, ref_a), name
for name, t in res.items():
    t = sorted(t)
    print(f"{name}: median {(t[1] + t[2]) / 2:.3f} ms  min {t[0]:.3f}  -> {pairs / ((t[1] + t[2]) / 2) * 1e3:.0f} pairs/s", flush=True)
