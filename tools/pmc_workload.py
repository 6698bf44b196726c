"""Workload for the rocprofv3 --pmc passes (profiles/): a few launches of every hot kernel at its benchmarked shape.
  the four GEMM instances of a ViT-H block at M = 65792 (batch 256), the fused in_proj + attention kernel and the
  stand-alone attention kernel at 256 images, LayerNorm, the 1M x 1024 scan (1 query) and the batched scan (16 queries).
Run as:  rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 tools/pmc_workload.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippomm_amd import _lib as L
from hippomm_amd.vector_ops import FeatureStore

lib = L.load()
B, T, D, H, MLP = 256, 257, 1280, 16, 5120
R = B * T
dev = "cuda"
REPS = 3
x = torch.randn(R, D, device=dev)
a = torch.randn(R, D, device=dev).to(torch.bfloat16)
big = torch.empty(R, MLP, dtype=torch.bfloat16, device=dev)
g1, b1 = torch.ones(D, device=dev), torch.zeros(D, device=dev)
wq = (torch.randn(3 * D, D, device=dev) * 0.02).to(torch.bfloat16); bq = torch.zeros(3 * D, device=dev)
wo = (torch.randn(D, D, device=dev) * 0.02).to(torch.bfloat16); bo = torch.zeros(D, device=dev)
w1 = (torch.randn(MLP, D, device=dev) * 0.02).to(torch.bfloat16); bb1 = torch.zeros(MLP, device=dev)
w2 = (torch.randn(D, MLP, device=dev) * 0.02).to(torch.bfloat16); bb2 = torch.zeros(D, device=dev)
qkv_cls = torch.zeros(B, 3 * D, dtype=torch.bfloat16, device=dev)
attn_out = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
S = L.stream_ptr
for _ in range(REPS):
    L.check(lib.hmm_op_layernorm_bf16(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), R, D, 1e-6, S()), "ln")
    L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), big.data_ptr(), R, 3 * D, D, 0, S()), "qkv")
    L.check(lib.hmm_op_attention_bf16(big.data_ptr(), attn_out.data_ptr(), B, T, H, D // H, None, None, S()), "attn")
    L.check(lib.hmm_op_qkv_attention_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), qkv_cls.data_ptr(), attn_out.data_ptr(), B, S()), "fused")
    L.check(lib.hmm_op_gemm_bf16(attn_out.data_ptr(), wo.data_ptr(), bo.data_ptr(), x.data_ptr(), R, D, D, 2, S()), "out")
    L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w1.data_ptr(), bb1.data_ptr(), big.data_ptr(), R, MLP, D, 1, S()), "fc1")
    L.check(lib.hmm_op_gemm_bf16(big.data_ptr(), w2.data_ptr(), bb2.data_ptr(), x.data_ptr(), R, D, MLP, 2, S()), "fc2")
torch.cuda.synchronize()
del x, a, big, attn_out
torch.cuda.empty_cache()
n = 1_000_000
rows = torch.empty(n, 1024, dtype=torch.float32, device=dev)
g = torch.Generator(device=dev).manual_seed(42)
for s in range(0, n, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device=dev)
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
store = FeatureStore(rows)
q = torch.randn(1024, device=dev)
q16 = torch.randn(16, 1024, device=dev)
store.build_shadow()
for _ in range(REPS):
    store.search_device(q, 32)
    store.search_multi_device(q16, 32)
    store.search_prefiltered_device(q, 32)
torch.cuda.synchronize()
print("pmc workload done")
