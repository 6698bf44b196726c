"""Joules per frame of the ViT-H forward at 256 frames, and where they go: board power (THIS GPU's hwmon sensor) sampled while the
whole two-chain forward loops, and while each kernel of a block loops alone at its full-batch shape.  The forward runs at the
board's power cap, so its time is its energy divided by ~1.36 kW: the account says which kernel owns how many joules.
usage: energy_account_probe.py [json_out]"""
import json
import sys
import threading
import time

from probe_common import load_probe, own_power_file

import torch

L, lib = load_probe()
from hippomm_amd.encoder import HipTower, synthetic_state_dict  # noqa: E402

PFILE = own_power_file()
if PFILE is None:
    raise SystemExit("no hwmon power sensor for this GPU")
samples, stop = [], False


def sampler():
    while not stop:
        try:
            samples.append((time.perf_counter(), int(open(PFILE).read()) / 1e6))
        except (OSError, ValueError):
            pass
        time.sleep(0.01)


def watts(t0, t1):
    v = [w for t, w in samples if t0 + 0.3 <= t <= t1 - 0.05]
    return sum(v) / len(v) if v else float("nan")


def loop(fn, secs, batch=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(batch):
            fn()
        torch.cuda.synchronize(); n += batch
    t1 = time.perf_counter()
    return (t1 - t0) / n * 1e3, watts(t0, t1)


th = threading.Thread(target=sampler); th.start()
time.sleep(1.5)
t_idle0 = time.perf_counter(); time.sleep(1.5); idle_w = watts(t_idle0 - 0.3, time.perf_counter() + 0.05)
B, T, D, H, MLP = 256, 257, 1280, 16, 5120
R = B * T
dev = "cuda"
res = {"power_sensor": PFILE, "idle_W": round(idle_w, 1), "frames": B}
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
frames = torch.randn(B, 3, 224, 224, device=dev); emb = torch.empty(B, 1024, device=dev)
for streams in (2, 1):
    tower.set_streams(streams)
    ms, w = loop(lambda: tower.forward_into(frames, emb), 5.0, batch=4)
    res[f"forward_{streams}_chain"] = {"ms": round(ms, 3), "board_W": round(w, 1), "J_per_forward": round(ms * w / 1e3, 2),
                                       "J_per_frame": round(ms * w / 1e3 / B, 4), "J_per_frame_above_idle": round(ms * (w - idle_w) / 1e3 / B, 4),
                                       "pJ_per_nominal_FLOP": round(ms * w / 1e3 / tower.flops(B) * 1e12, 3)}
tower.set_streams(2)
del tower
torch.cuda.empty_cache()
x = torch.randn(R, D, device=dev)
a = torch.randn(R, D, device=dev).to(torch.bfloat16)
big = torch.empty(R, MLP, dtype=torch.bfloat16, device=dev)
g1, b1 = torch.ones(D, device=dev), torch.zeros(D, device=dev)
wq = (torch.randn(3 * D, D, device=dev) * 0.02).to(torch.bfloat16); bq = torch.zeros(3 * D, device=dev)
wo = (torch.randn(D, D, device=dev) * 0.02).to(torch.bfloat16); bo = torch.zeros(D, device=dev)
w1 = (torch.randn(MLP, D, device=dev) * 0.02).to(torch.bfloat16); bb1 = torch.zeros(MLP, device=dev)
w2 = (torch.randn(D, MLP, device=dev) * 0.02).to(torch.bfloat16); bb2 = torch.zeros(D, device=dev)
qkv_cls = torch.zeros(B, 3 * D, dtype=torch.bfloat16, device=dev)
S = L.stream_ptr
gemm = lambda A, W, Bv, Cc, N, K, epi: L.check(lib.hmm_op_gemm_bf16(A.data_ptr(), W.data_ptr(), Bv.data_ptr(), Cc.data_ptr(), R, N, K, epi, S()), "gemm")
# realistic operand statistics: `a` is a LayerNorm output, `big` a GELU output when fc2 reads it
L.check(lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S()), "ln")
kernels = [
    ("layernorm x2", 2, None, lambda: L.check(lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S()), "ln")),
    ("in_proj+attention (fused)", 1, 2.0 * R * 3 * D * D + 4.0 * B * H * T * T * (D // H),
     lambda: L.check(lib.hmm_op_qkv_attention_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), qkv_cls.data_ptr(), big.data_ptr(), B, S()), "fused")),
    ("out_proj+residual", 1, 2.0 * R * D * D, lambda: gemm(big, wo, bo, x, D, D, 2)),
    ("fc1+gelu", 1, 2.0 * R * MLP * D, lambda: gemm(a, w1, bb1, big, MLP, D, 1)),
    ("fc2+residual", 1, 2.0 * R * D * MLP, lambda: gemm(big, w2, bb2, x, D, MLP, 2)),
]
acct, tot_j, tot_ms = [], 0.0, 0.0
for name, per_block, flops, fn in kernels:
    if name.startswith("fc2"):
        gemm(a, w1, bb1, big, MLP, D, 1)                       # big = a GELU output
    if name.startswith("out_proj"):
        x.normal_()
    ms, w = loop(fn, 2.5, batch=20)
    j = ms * w / 1e3 * per_block
    rec = {"kernel": name, "launches_per_block": per_block, "ms_per_launch_alone": round(ms, 4), "board_W": round(w, 1),
           "J_per_block": round(j, 4), "J_per_block_above_idle": round(ms * (w - idle_w) / 1e3 * per_block, 4)}
    if flops:
        rec["TFLOPs"] = round(flops / ms / 1e9, 0)
        rec["pJ_per_FLOP_all_in"] = round(ms * w / 1e3 / flops * 1e12, 3)
    acct.append(rec); tot_j += j; tot_ms += ms * per_block
    print(rec, flush=True)
    x.normal_()
stop = True; th.join()
res["per_block_alone"] = acct
res["sum_of_kernels_alone_x32_blocks"] = {"J": round(tot_j * 32, 1), "ms": round(tot_ms * 32, 2), "J_per_frame": round(tot_j * 32 / B, 4)}
f2 = res["forward_2_chain"]
res["reading"] = ("forward J = board W x ms; it equals (within the sensor's few %) the sum of the kernels' own joules: running two chains "
                  "does not remove energy, it removes idle time (static power x time), which is why the two-chain forward is "
                  f"{round((1 - f2['ms'] / res['forward_1_chain']['ms']) * 100, 1)} % faster at about the same joules per frame")
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
