"""Attention core timing with the built-in ablations (hmm_dev_set_attn_reverse bits: 1 reverse image order,
2 skip compute, 4 skip K/V global loads, 8 XCD-aware block order)."""
import sys
sys.path.insert(0, ".")
import torch
from hippomm_amd import _lib as L
lib = L.load()
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tower = sys.argv[2] if len(sys.argv) > 2 else "vision"
T, H, DH = {"vision": (257, 16, 80), "audio": (229, 12, 64), "text": (77, 16, 64)}[tower]
D = H * DH
qkv = (torch.randn(n_img * T, 3 * D, device="cuda") * 0.5).bfloat16()
out = torch.empty(n_img * T, D, dtype=torch.bfloat16, device="cuda")
bk = torch.randn(D, device="cuda") if tower == "audio" else None
bv = torch.randn(D, device="cuda") if tower == "audio" else None
st = L.stream_ptr()


def run():
    return lib.hmm_op_attention_bf16(qkv.data_ptr(), out.data_ptr(), n_img, T, H, DH, bk.data_ptr() if bk is not None else None,
                                     bv.data_ptr() if bv is not None else None, st)


def timeit(it=20):
    for _ in range(3): L.check(run(), "attn")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


flops = 4.0 * T * T * DH * H * n_img
for mode, name in [(1, "full"), (0, "full, forward image order"), (9, "full, XCD-aware"), (8, "full, XCD-aware forward"),
                   (3, "no compute (stage K/V only)"), (11, "no compute, XCD-aware"), (5, "no global K/V loads"), (7, "neither")]:
    lib.hmm_dev_set_attn_reverse(mode)
    t = timeit()
    print(f"{tower} n_img={n_img} {name:32s} {t:8.1f} us   {flops / t / 1e6:7.1f} TFLOP/s", flush=True)
lib.hmm_dev_set_attn_reverse(9)

# reference point: torch SDPA (vendor flash attention) on the same problem, q/k/v already split and head-major
import torch.nn.functional as F
q, k, v = (torch.randn(n_img, H, T, DH, device="cuda").bfloat16() * 0.5 for _ in range(3))
for _ in range(3): F.scaled_dot_product_attention(q, k, v)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): F.scaled_dot_product_attention(q, k, v)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print(f"{tower} n_img={n_img} torch SDPA (contiguous head-major q/k/v, no packing/unpacking): {t:8.1f} us   {flops / t / 1e6:7.1f} TFLOP/s")
