"""A/B in one process: ViT-H forward at 256 frames with LayerNorm folded into the GEMMs (default) vs as its own kernel."""
import sys
from probe_common import ROOT, event_ms  # noqa: F401  (puts the repo root on sys.path)
import torch
from hippomm_amd.encoder import HipTower, synthetic_state_dict

tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
for B in (256, 32):
    x = torch.randn(B, 3, 224, 224, device="cuda"); out = torch.empty(B, 1024, device="cuda")
    for rnd in range(3):
        for on in (1, 0):
            tower.set_folded_layernorm(bool(on))
            ms = event_ms(lambda: tower.forward_into(x, out), 5, warmup=2)
            print(f"B={B} folded_layernorm={on}: {ms:.2f} ms  {B / ms * 1e3:.0f} img/s", flush=True)
tower.set_folded_layernorm(True)
