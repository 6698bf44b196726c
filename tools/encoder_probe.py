"""Time the full ViT-H (32 blocks) forward at B=256 for each GEMM variant."""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from hippomm_amd import _lib as L
from hippomm_amd.encoder import HipTower, synthetic_state_dict
lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tower_name = sys.argv[2] if len(sys.argv) > 2 else "vision"
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2]
sd = synthetic_state_dict((tower_name,))
tower = HipTower(tower_name, sd)
del sd
if tower_name == "text":
    x = torch.randint(1, 49000, (B, 77), device="cuda"); x[:, 20] = 49407; x[:, 21:] = 0
else:
    shape = (B, 3, 224, 224) if tower_name == "vision" else (B, 3, 1, 128, 204)
    x = torch.randn(*shape, device="cuda")
out = torch.empty(B, 1024, device="cuda")
import itertools
streams = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2]
lnb = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
tgs = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 else [-1]
for v, ns, lb, tg in itertools.product(variants, streams, lnb, tgs):
    lib.hmm_dev_set_gemm_tile_group(tg)
    lib.hmm_dev_set_gemm_variant(v)
    lib.hmm_dev_set_encoder_streams(ns)
    lib.hmm_dev_set_ln_max_blocks(lb)
    for _ in range(2): tower.forward_into(x, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it): tower.forward_into(x, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = tower.flops(B)
    print(f"{tower_name} B={B} gemm_variant={v} streams={ns} ln_blocks={lb} tile_group={tg}: {ms:.2f} ms/forward  {B/ms*1e3:.1f} samples/s  "
          f"{fl/ms/1e9:.0f} TFLOP/s ({fl/ms/1e9/2500*100:.1f}% of 2.5 PF)", flush=True)
print("out finite:", bool(torch.isfinite(out).all()), "norm", float(out.norm(dim=1).mean()))
