#!/bin/bash
# dispatcher audit follow-ups: 64 x 128 ring tiles past 410 tiles of 64 x 64; ping-pong for the qkv GEMM from 88 tiles on
mkdir -p gpurun_out
timeout 500 python tools/knob_ab_probe.py g_gemm_rect64_min_t64 0 410 gpurun_out/rect64_ab_vision.json vision:4,5,6,7,9,12 2>&1 | grep ratio
timeout 400 python tools/knob_ab_probe.py g_gemm_rect64_min_t64 0 410 gpurun_out/rect64_ab_text.json text:20,22,24,26,28 2>&1 | grep ratio
timeout 300 python tools/knob_ab_probe.py g_gemm_rect64_min_t64 0 410 gpurun_out/rect64_ab_audio.json audio:3,4 2>&1 | grep ratio
timeout 500 python tools/knob_ab_probe.py g_gemm_pp_bias_tiles 0 88 gpurun_out/ppbias_ab_vision.json vision:4,5,6,7,8 2>&1 | grep ratio
timeout 400 python tools/knob_ab_probe.py g_gemm_pp_bias_tiles 0 88 gpurun_out/ppbias_ab_text.json text:28,32,40,48 2>&1 | grep ratio
timeout 300 python tools/knob_ab_probe.py g_gemm_pp_bias_tiles 0 88 gpurun_out/ppbias_ab_audio.json audio:3,4,5,6,8 2>&1 | grep ratio
