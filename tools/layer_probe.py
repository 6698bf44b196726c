"""Time every kernel of one ViT-H transformer block in sequence context (B=256) with HIP events,
several rounds, optional A/B of a tuning flag inside one process.
usage: layer_probe.py [flag=val,val ...]   flags: attn_reverse, gemm_variant"""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from hippomm_amd import _lib as L
lib = L.load()
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
steps = [
    ("ln1", lambda: lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S())),
    ("qkv", lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), big.data_ptr(), R, 3 * D, D, 0, S())),
    ("attn", lambda: lib.hmm_op_attention_bf16(big.data_ptr(), a.data_ptr(), B, T, H, D // H, None, None, S())),
    ("out", lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), R, D, D, 2, S())),
    ("ln2", lambda: lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S())),
    ("fc1", lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), w1.data_ptr(), bb1.data_ptr(), big.data_ptr(), R, MLP, D, 1, S())),
    ("fc2", lambda: lib.hmm_op_gemm_bf16(big.data_ptr(), w2.data_ptr(), bb2.data_ptr(), x.data_ptr(), R, D, MLP, 2, S())),
]
setters = {"attn_reverse": lib.hmm_dev_set_attn_reverse, "gemm_variant": lib.hmm_dev_set_gemm_variant}
configs = [{}]
for arg in sys.argv[1:]:
    k, vs = arg.split("=")
    configs = [dict(c, **{k: int(v)}) for c in configs for v in vs.split(",")]

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

results = {i: [] for i in range(len(configs))}
for rnd in range(3):
    for i, cfg in enumerate(configs):
        for k, v in cfg.items(): setters[k](v)
        x.normal_()
        run_layers(2)
        results[i].append(run_layers(8))
for i, cfg in enumerate(configs):
    med = {k: sorted(r[k] for r in results[i])[1] for k in results[i][0]}
    tot = sum(med.values())
    print(cfg, " ".join(f"{k}={v:.0f}" for k, v in med.items()), f"| layer={tot:.0f} us  x32={tot*32/1e3:.1f} ms", flush=True)
