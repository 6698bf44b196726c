"""In-situ A/B of tuning knobs of the probe build (tools/libhippomm_probe.so), in ONE process, interleaved rounds:
  * every kernel of one ViT-H transformer block timed in sequence context (B=256, HIP events around each launch),
  * the whole 32-block forward at B=256 (two streams), median of the rounds.
usage: layer_probe.py [knob=v0,v1,...] ... [--no-tower] [--json out.json]
       knobs are the hmm_probe_set_<knob> setters of the probe build (e.g. g_gemm_dephase_units=0,2,4)."""
import itertools
import json
import sys

from probe_common import load_probe, setter, event_ms

import torch

L, lib = load_probe()
tower = None
args = [a for a in sys.argv[1:] if not a.startswith("--")]
no_tower = "--no-tower" in sys.argv
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
args = [a for a in args if a != json_out]
knobs = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in args]
configs = [dict(zip([k for k, _ in knobs], vals)) for vals in itertools.product(*[v for _, v in knobs])] or [{}]

B, T, D, H, MLP = 256, 257, 1280, 16, 5120
R = B * T
dev = "cuda"
x = torch.randn(R, D, device=dev)
a = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
big = torch.empty(R, MLP, dtype=torch.bfloat16, device=dev)
g1, b1 = torch.ones(D, device=dev), torch.zeros(D, device=dev)
wq = (torch.randn(3 * D, D, device=dev) * 0.02).to(torch.bfloat16); bq = torch.zeros(3 * D, device=dev)
wo = (torch.randn(D, D, device=dev) * 0.02).to(torch.bfloat16); bo = torch.zeros(D, device=dev)
w1 = (torch.randn(MLP, D, device=dev) * 0.02).to(torch.bfloat16); bb1 = torch.zeros(MLP, device=dev)
w2 = (torch.randn(D, MLP, device=dev) * 0.02).to(torch.bfloat16); bb2 = torch.zeros(D, device=dev)
S = L.stream_ptr
cls_rows = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
qkv_cls = torch.empty(B, 3 * D, dtype=torch.bfloat16, device=dev)


def cls_proj():
    cls_rows.copy_(a.view(B, T, D)[:, 0])
    return lib.hmm_op_gemm_bf16(cls_rows.data_ptr(), wq.data_ptr(), bq.data_ptr(), qkv_cls.data_ptr(), B, 3 * D, D, 0, S())


steps_fused = [
    ("ln1", lambda: lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S())),
    ("clsqkv", cls_proj),
    ("qkvattn", lambda: lib.hmm_op_qkv_attention_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), qkv_cls.data_ptr(), big.data_ptr(), B, S())),
    ("out", lambda: lib.hmm_op_gemm_bf16(big.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), R, D, D, 2, S())),
    ("ln2", lambda: lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S())),
    ("fc1", lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), w1.data_ptr(), bb1.data_ptr(), big.data_ptr(), R, MLP, D, 1, S())),
    ("fc2", lambda: lib.hmm_op_gemm_bf16(big.data_ptr(), w2.data_ptr(), bb2.data_ptr(), x.data_ptr(), R, D, MLP, 2, S())),
]
steps_plain = [
    ("ln1", lambda: lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S())),
    ("qkv", lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), big.data_ptr(), R, 3 * D, D, 0, S())),
    ("attn", lambda: lib.hmm_op_attention_bf16(big.data_ptr(), a.data_ptr(), B, T, H, D // H, None, None, S())),
    ("out", lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), R, D, D, 2, S())),
    ("ln2", lambda: lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S())),
    ("fc1", lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), w1.data_ptr(), bb1.data_ptr(), big.data_ptr(), R, MLP, D, 1, S())),
    ("fc2", lambda: lib.hmm_op_gemm_bf16(big.data_ptr(), w2.data_ptr(), bb2.data_ptr(), x.data_ptr(), R, D, MLP, 2, S())),
]


steps = steps_plain


def apply(cfg):
    global steps
    for k, v in cfg.items():
        if k == "fused":                     # pseudo-knob: fused in_proj + attention kernel vs GEMM + attention
            steps = steps_fused if v else steps_plain
            if tower is not None:
                tower.set_fused_attention(bool(v))
        elif k == "streams":                 # pseudo-knob: hmm_encoder_set_streams (1 = one chain, 2 = two half-batches)
            if tower is not None:
                tower.set_streams(v)
        else:
            setter(lib, k)(v)


def run_layers(n_layers):
    evs = []
    for _ in range(n_layers):
        for name, fn in steps:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); L.check(fn(), name); e1.record()
            evs.append((name, e0, e1))
    torch.cuda.synchronize()
    acc = {}
    for name, e0, e1 in evs:
        acc.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items()}


frames = emb = None
if not no_tower:
    from hippomm_amd.encoder import HipTower, synthetic_state_dict
    sd = synthetic_state_dict(("vision",), seed=1234)
    tower = HipTower("vision", sd)
    del sd
    torch.cuda.empty_cache()
    frames = torch.randn(B, 3, 224, 224, device=dev)
    emb = torch.empty(B, 1024, device=dev)

ROUNDS = 3
layer_res = {i: [] for i in range(len(configs))}
tower_res = {i: [] for i in range(len(configs))}
for rnd in range(ROUNDS):
    for i, cfg in enumerate(configs):
        apply(cfg)
        x.normal_()
        run_layers(2)
        layer_res[i].append(run_layers(6))
        if tower is not None:
            tower_res[i].append(event_ms(lambda: tower.forward_into(frames, emb), 4, warmup=1))
out = []
for i, cfg in enumerate(configs):
    med = {k: sorted(r[k] for r in layer_res[i])[ROUNDS // 2] for k in layer_res[i][0]}
    tot = sum(med.values())
    rec = {"cfg": cfg, "layer_us": {k: round(v) for k, v in med.items()}, "layer_total_us": round(tot)}
    if tower is not None:
        t = sorted(tower_res[i])
        rec["forward_ms_median"] = round(t[ROUNDS // 2], 2)
        rec["forward_ms_min"] = round(t[0], 2)
        rec["img_per_s"] = round(B / t[ROUNDS // 2] * 1e3)
    out.append(rec)
    print(rec, flush=True)
if json_out:
    json.dump(out, open(json_out, "w"), indent=1)
