"""Where does a workgroup of the fused in_proj + attention kernel spend its time?  In-kernel stamps of the probe build: start /
projection loop done / Q, K, V images written / main query tiles done / end, and the shader clock over the projection and the
attention phase.  256 frames, vision geometry.  usage: fused_stamp_probe.py [json_out]"""
import ctypes as C
import json
import sys

from probe_common import load_probe

import numpy as np
import torch

L, lib = load_probe()
lib.hmm_probe_set_fused_stamps.restype = None
lib.hmm_probe_set_fused_stamps.argtypes = [C.c_void_p]
B, T, D, H = 256, 257, 1280, 16
a = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
w = (torch.randn(3 * D, D, device="cuda") * 0.02).to(torch.bfloat16)
bias = torch.zeros(3 * D, device="cuda")
qkv_cls = torch.randn(B, 3 * D, device="cuda").to(torch.bfloat16)
out = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
n_wg = B * H
stamps = torch.zeros(n_wg * 16, dtype=torch.int64, device="cuda")


def run():
    L.check(lib.hmm_op_qkv_attention_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), qkv_cls.data_ptr(), out.data_ptr(), B,
                                          L.stream_ptr()), "fused")


lib.hmm_probe_set_fused_stamps(None)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
plain = e0.elapsed_time(e1)
lib.hmm_probe_set_fused_stamps(stamps.data_ptr())
e0.record(); run(); e1.record(); torch.cuda.synchronize()
stamped = e0.elapsed_time(e1)
lib.hmm_probe_set_fused_stamps(None)
s = stamps.cpu().numpy().reshape(n_wg, 16).astype(np.float64)
us = lambda i, j: (s[:, j] - s[:, i]) / 100.0
pct = lambda v: [round(float(np.percentile(v, q)), 2) for q in (10, 50, 90)]
clk_proj = (s[:, 7] - s[:, 6]) / ((s[:, 1] - s[:, 0]) * 10.0)
clk_attn = (s[:, 8] - s[:, 7]) / ((s[:, 4] - s[:, 1]) * 10.0)
rec = {"frames": B, "workgroups": n_wg, "kernel_ms_plain": round(plain, 4), "kernel_ms_stamped": round(stamped, 4),
       "projection_fill_plus_loop_us": pct(us(0, 1)), "accumulators_to_qkv_images_us": pct(us(1, 2)),
       "main_query_tiles_us": pct(us(2, 3)), "cooperative_257th_query_and_stores_us": pct(us(3, 4)),
       "attention_phase_total_us": pct(us(1, 4)), "workgroup_total_us": pct(us(0, 4)),
       "clock_GHz_projection": pct(clk_proj), "clock_GHz_attention_phase": pct(clk_attn),
       "mfma_floor_us_attention_at_that_clock": round(891 * 32 / 4 / (float(np.median(clk_attn)) * 1e3), 2),
       "rounds_of_256_cus": n_wg / 256}
print(json.dumps(rec, indent=1))
if len(sys.argv) > 1:
    json.dump(rec, open(sys.argv[1], "w"), indent=1)
