"""A/B of probe-build GEMM knobs on the four ViT-H shapes at M=65792, interleaved rounds, plus a bitwise check of every
variant against variant 0.   usage: gemm_ab_probe.py knob=v0,v1,..."""
import sys
from probe_common import load_probe, setter, event_ms
import torch

L, lib = load_probe()
knob, vals = sys.argv[1].split("=")
vals = [int(v) for v in vals.split(",")]
M = 65792
shapes = [("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2)]
for name, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    c0 = torch.randn(M, N, device="cuda").to(torch.float32 if epi == 2 else torch.bfloat16)
    outs, times = {}, {v: [] for v in vals}
    for v in vals:
        setter(lib, knob)(v)
        c = c0.clone()
        L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, L.stream_ptr()), "gemm")
        outs[v] = c
    same = {v: bool(torch.equal(outs[v], outs[vals[0]])) for v in vals}
    c = c0.clone()
    for rnd in range(5):
        for v in vals:
            setter(lib, knob)(v)
            times[v].append(event_ms(lambda: L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(),
                                                                           M, N, K, epi, L.stream_ptr()), "gemm"), 10, warmup=2))
    for v in vals:
        t = sorted(times[v])[2]
        print(f"{name:4s} {knob}={v}: {t*1e3:7.1f} us  {2*M*N*K/t/1e9:7.1f} TFLOP/s  bitwise_equal_to_first={same[v]}", flush=True)
    del a, w, c0, c, outs
