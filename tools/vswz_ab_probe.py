"""One build of the library per process (HMM_PROBE_LIB): the fused in_proj + attention kernels and the forward.  Run twice, with
the swizzled-V build and the plain one, in the same box session and compare.  usage: vswz_ab_probe.py <tag>"""
import sys
from probe_common import load_probe, event_ms
import torch
L, lib = load_probe()
tag = sys.argv[1] if len(sys.argv) > 1 else "?"
B, T, D = 256, 257, 1280
a = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
wq = (torch.randn(3 * D, D, device="cuda") * 0.02).to(torch.bfloat16); bq = torch.zeros(3 * D, device="cuda")
qkv_cls = torch.zeros(B, 3 * D, dtype=torch.bfloat16, device="cuda")
out = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
S = L.stream_ptr
fused = lambda: L.check(lib.hmm_op_qkv_attention_bf16(a.data_ptr(), wq.data_ptr(), bq.data_ptr(), qkv_cls.data_ptr(), out.data_ptr(), B, S()), "fused")
Ta, Da, Ca = 229, 768, 384
aa = torch.randn(Ca * Ta, Da, device="cuda").to(torch.bfloat16)
wa = (torch.randn(3 * Da, Da, device="cuda") * 0.02).to(torch.bfloat16); ba = torch.zeros(3 * Da, device="cuda")
bk, bv = torch.randn(Da, device="cuda"), torch.randn(Da, device="cuda")
oa = torch.empty(Ca * Ta, Da, dtype=torch.bfloat16, device="cuda")
fused_a = lambda: L.check(lib.hmm_op_qkv_attention_audio_bf16(aa.data_ptr(), wa.data_ptr(), ba.data_ptr(), bk.data_ptr(), bv.data_ptr(), oa.data_ptr(), Ca, S()), "fused audio")
qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
attn = lambda: L.check(lib.hmm_op_attention_bf16(qkv.data_ptr(), out.data_ptr(), B, T, 16, 80, None, None, S()), "attn")
for rnd in range(3):
    print(f"[{tag}] fused vision {event_ms(fused, 10, warmup=3) * 1e3:7.1f} us   fused audio {event_ms(fused_a, 10, warmup=3) * 1e3:7.1f} us   "
          f"attention alone {event_ms(attn, 10, warmup=3) * 1e3:7.1f} us", flush=True)
from hippomm_amd.encoder import HipTower, synthetic_state_dict
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
x = torch.randn(256, 3, 224, 224, device="cuda"); e = torch.empty(256, 1024, device="cuda")
for rnd in range(3):
    print(f"[{tag}] forward {event_ms(lambda: tower.forward_into(x, e), 5, warmup=2):.3f} ms", flush=True)
