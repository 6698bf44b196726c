"""cfg 5 fingerprint mismatch hunt: the same synthetic video frames encoded in 256-frame batches that start at different
offsets (as N = 1 and N = 2 shard them), folded LayerNorm on / off, twice each (run-to-run)."""
import sys
from probe_common import ROOT  # noqa: F401
import torch
import bench
from hippomm_amd.encoder import HipTower, synthetic_state_dict

tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
torch.cuda.empty_cache()
lo, hi = 1536, 2400
frames = bench.synthetic_frames(lo, hi, "cuda")

def encode(start, stop, step=256):
    out = torch.empty(stop - start, 1024, device="cuda")
    for s in range(start, stop, step):
        e = min(s + step, stop)
        tower.forward_into(frames[s - lo:e - lo], out[s - start:e - start])
    torch.cuda.synchronize()
    return out

for fold in (0, 1):
    tower.set_folded_layernorm(bool(fold))
    a1 = encode(1536, 2304)                       # batches [1536,1792) [1792,2048) [2048,2304)
    a2 = encode(1536, 2304)
    b1 = encode(1800, 2400)                       # batches [1800,2056) [2056,2312) [2312,2400)
    b2 = encode(1800, 2400)
    c = encode(1800, 2400, step=32)
    print(f"fold {fold}: run-to-run a {torch.equal(a1, a2)} b {torch.equal(b1, b2)}", flush=True)
    ov_a, ov_b, ov_c = a1[1800 - 1536:], b1[:2304 - 1800], c[:2304 - 1800]
    bad = (ov_a != ov_b).any(dim=1).nonzero().flatten()
    print(f"fold {fold}: frames 1800..2303, offset-0 batches vs offset-1800 batches: {bad.numel()} rows differ; first {[(1800 + int(i)) for i in bad[:12]]}", flush=True)
    bad = (ov_b != ov_c).any(dim=1).nonzero().flatten()
    print(f"fold {fold}: offset-1800 256-batches vs 32-batches: {bad.numel()} rows differ; first {[(1800 + int(i)) for i in bad[:12]]}", flush=True)
    if bad.numel():
        i = int(bad[0]); d = (ov_b[i] - ov_c[i]).abs()
        print("   max abs diff in that row", d.max().item(), "n elems", int((d > 0).sum()))
