"""Where does bitwise batch invariance of the vision tower break?  Same frames through different batch splits, folded
LayerNorm on / off, depth 4 and 32.  usage: batch_invariance_probe.py"""
import sys
from probe_common import ROOT  # noqa: F401
import torch
from hippomm_amd.encoder import HipTower, synthetic_state_dict

for depth in (4, 32):
    tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234, depth={"vision": depth}), depth=depth)
    x = torch.randn(272, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    for fold in (1, 0):
        tower.set_folded_layernorm(bool(fold))
        ref = torch.cat([tower(x[i:i + 1]) for i in range(0, 24)])          # one frame at a time
        for split in ([24], [8, 16], [16, 8], [8, 8, 8], [4, 20]):
            outs, s = [], 0
            for n in split:
                outs.append(tower(x[s:s + n])); s += n
            got = torch.cat(outs)
            bad = (got != ref).any(dim=1).nonzero().flatten().tolist()
            print(f"depth {depth} fold {fold} split {split}: rows differing from the one-by-one result: {bad}", flush=True)
        a = torch.cat([tower(x[:256]), tower(x[256:272])])
        b = torch.cat([tower(x[:8]), tower(x[8:264]), tower(x[264:272])])
        c = tower(x, max_batch=64)
        print(f"depth {depth} fold {fold}: 256+16 vs 8+256+8 differ in {(a != b).any(dim=1).sum().item()} rows; "
              f"vs 64-chunks {(a != c).any(dim=1).sum().item()} rows", flush=True)
        print("   rows:", (a != b).any(dim=1).nonzero().flatten().tolist()[:40], flush=True)
    del tower
    torch.cuda.empty_cache()
