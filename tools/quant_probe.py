"""What does the partly filled last round of the half-batch N = 1280 GEMMs (645 tiles = 2.52 rounds of 256 CUs) cost in the
two-stream forward?  Timing-only experiment: with g_gemm_trunc_rounds the launch drops the 133 tiles of the third round (20.6 %
of those GEMMs' work; results wrong).  If the other chain already fills the idle CUs, the forward shrinks by about the work
removed (fc2 + out-proj are ~38 % of a block: -7.8 %); if the third round costs a whole round, by ~-12.7 %."""
from probe_common import load_probe, setter, event_ms
import torch
L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
torch.cuda.empty_cache()
x = torch.randn(256, 3, 224, 224, device="cuda"); out = torch.empty(256, 1024, device="cuda")
times = {0: [], 1: []}
for rnd in range(4):
    for k in (0, 1):
        setter(lib, "g_gemm_trunc_rounds")(k)
        times[k].append(event_ms(lambda: tower.forward_into(x, out), 4, warmup=2))
setter(lib, "g_gemm_trunc_rounds")(0)
m = {k: sorted(v)[1] for k, v in times.items()}
print(f"forward, all tiles: {m[0]:.3f} ms; third rounds of the N = 1280 half-batch GEMMs dropped: {m[1]:.3f} ms ({(m[1] / m[0] - 1) * 100:+.1f} %)", flush=True)
# the same GEMMs alone: 645 vs 512 tiles
M, D, H = 128 * 257, 1280, 5120
for name, K in (("out-proj", D), ("fc2", H)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(D, K, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.zeros(D, device="cuda"); c = torch.zeros(M, D, device="cuda")
    run = lambda: L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, D, K, 2, 3, L.stream_ptr()), "g")
    r = {}
    for k in (0, 1):
        setter(lib, "g_gemm_trunc_rounds")(k)
        r[k] = event_ms(run, 10, warmup=3) * 1e3
    setter(lib, "g_gemm_trunc_rounds")(0)
    print(f"{name} half batch alone: 645 tiles {r[0]:.1f} us, 512 tiles {r[1]:.1f} us ({(r[1] / r[0] - 1) * 100:+.1f} %; work -20.6 %)", flush=True)
