"""Mid-size vision batches (16 .. 64 frames): chains (1 / 2) x the tile count below which a launch leaves the 256x256 ping-pong
kernel for the small-tile kernels.  usage: mid_batch_probe.py"""
import sys
import time
from probe_common import load_probe, setter
import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict
set_min = setter(lib, "g_enc_split_min")
set_small = setter(lib, "g_enc_two_chain_small_tiles")


def wall_ms(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


for B in ([int(a) for a in sys.argv[1:]] or (16, 20, 24, 28, 32, 40, 48, 56, 64, 96, 128, 256)):
    torch.manual_seed(B)
    x = torch.randn(B, 3, 224, 224, device="cuda")
    ref = None
    line = f"vision B={B}:"
    for chains, m in (("1ch", 1 << 30), ("2ch", 2)):
        set_min(m)
        tower = HipTower("vision", synthetic_state_dict(("vision",), seed=99))
        out = torch.empty(B, 1024, device="cuda")
        for small in ((128, 256) if chains == '1ch' else (128, 64, 32)):
            set_small(small)
            ms = wall_ms(lambda: tower.forward_into(x, out))
            if ref is None:
                ref = out.clone()
            line += f"  {chains}/small<{small}: {ms:.2f}{'' if torch.equal(ref, out) else ' BITS DIFFER'}"
        del tower
    print(line, flush=True)
set_min(16); set_small(64)
