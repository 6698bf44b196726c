"""Kernel-level cost of the folded-LayerNorm pieces at the benchmark shape (65 792 x 1280), next to what they replace."""
import ctypes as C
from probe_common import load_probe, event_ms
import torch
L, lib = load_probe()
B, T, D, MLP = 256, 257, 1280, 5120
R = B * T
S = L.stream_ptr
x = torch.randn(R, D, device="cuda")
xb = x.to(torch.bfloat16)
a = torch.empty(R, D, dtype=torch.bfloat16, device="cuda")
big = torch.empty(R, MLP, dtype=torch.bfloat16, device="cuda")
g1, b1 = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
rs = torch.empty(R, 2, device="cuda")
part = torch.empty(R, D // 64, 2, device="cuda")
wo = (torch.randn(D, D, device="cuda") * 0.02).to(torch.bfloat16); bo = torch.zeros(D, device="cuda")
w1 = (torch.randn(MLP, D, device="cuda") * 0.02).to(torch.bfloat16); bb1 = torch.zeros(MLP, device="cuda"); c1 = torch.randn(MLP, device="cuda")
w2 = (torch.randn(D, MLP, device="cuda") * 0.02).to(torch.bfloat16); bb2 = torch.zeros(D, device="cuda")
wq = (torch.randn(3 * D, D, device="cuda") * 0.02).to(torch.bfloat16); bq = torch.zeros(3 * D, device="cuda"); cq = torch.randn(3 * D, device="cuda")
qkv_cls = torch.zeros(B, 3 * D, dtype=torch.bfloat16, device="cuda")
L.check(lib.hmm_op_rowstat_bf16(xb.data_ptr(), rs.data_ptr(), R, D, 1e-6, S()), "rs")
tests = {
 "layernorm (replaced)": lambda: lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S()),
 "rowstat": lambda: lib.hmm_op_rowstat_bf16(xb.data_ptr(), rs.data_ptr(), R, D, 1e-6, S()),
 "rowstat finalize": lambda: lib.hmm_op_rowstat_finalize(part.data_ptr(), rs.data_ptr(), R, D, 1e-6, S()),
 "out-proj resid": lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), R, D, D, 2, S()),
 "out-proj resid+xb": lambda: lib.hmm_op_gemm_bf16_resid_xb(a.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), xb.data_ptr(), part.data_ptr(), R, D, D, -1, S()),
 "fc2 resid": lambda: lib.hmm_op_gemm_bf16(big.data_ptr(), w2.data_ptr(), bb2.data_ptr(), x.data_ptr(), R, D, MLP, 2, S()),
 "fc2 resid+xb": lambda: lib.hmm_op_gemm_bf16_resid_xb(big.data_ptr(), w2.data_ptr(), bb2.data_ptr(), x.data_ptr(), xb.data_ptr(), part.data_ptr(), R, D, MLP, -1, S()),
 "fc1 bias+gelu": lambda: lib.hmm_op_gemm_bf16(a.data_ptr(), w1.data_ptr(), bb1.data_ptr(), big.data_ptr(), R, MLP, D, 1, S()),
 "fc1 ln+gelu": lambda: lib.hmm_op_gemm_bf16_ln(xb.data_ptr(), w1.data_ptr(), bb1.data_ptr(), big.data_ptr(), R, MLP, D, 1, rs.data_ptr(), 1, c1.data_ptr(), -1, S()),
 "fused qkv+attn": lambda: lib.hmm_op_qkv_attention_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), qkv_cls.data_ptr(), big.data_ptr(), B, S()),
 "fused qkv+attn ln": lambda: lib.hmm_op_qkv_attention_ln_bf16(xb.data_ptr(), wq.data_ptr(), bq.data_ptr(), qkv_cls.data_ptr(), big.data_ptr(), B, rs.data_ptr(), cq.data_ptr(), S()),
}
for rnd in range(2):
    for name, fn in tests.items():
        ms = event_ms(lambda: L.check(fn(), name), 10, warmup=2)
        print(f"{name:22s} {ms * 1e3:7.1f} us", flush=True)
