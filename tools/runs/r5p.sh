#!/bin/bash
# ring tail peel (gemm_bf16.hip g_gemm_ring_peel_rows): forwards A/B
mkdir -p gpurun_out
timeout 600 python tools/knob_ab_probe.py g_gemm_ring_peel_rows 0 16 gpurun_out/ring_peel_ab_vision.json vision:2,3,4,5,6,7,8,9,10,12,16,24 > gpurun_out/ring_peel_ab_vision.log 2>&1
timeout 300 python tools/knob_ab_probe.py g_gemm_ring_peel_rows 0 16 gpurun_out/ring_peel_ab_audio.json audio:2,3,4,5,6,8,12 > gpurun_out/ring_peel_ab_audio.log 2>&1
timeout 300 python tools/knob_ab_probe.py g_gemm_ring_peel_rows 0 16 gpurun_out/ring_peel_ab_text.json text:10,14,16,20,27,32,40 > gpurun_out/ring_peel_ab_text.log 2>&1
grep -h ratio gpurun_out/ring_peel_ab_*.log | cut -c1-200
