#!/bin/bash
# the staggered eight-wave ring loop: isolated GEMMs, then forwards A/B
mkdir -p gpurun_out
python tools/stagger_probe.py gpurun_out/stagger_probe.json > gpurun_out/stagger_probe.log 2>&1
python tools/knob_ab_probe.py g_gemm_stagger 0 1 gpurun_out/stagger_ab_vision.json vision:1,3,4,6,8,12,13,16,20,24,32,48 > gpurun_out/stagger_ab_vision.log 2>&1
python tools/knob_ab_probe.py g_gemm_stagger 0 2 gpurun_out/stagger_ab2_vision.json vision:1,2,3 > gpurun_out/stagger_ab2_vision.log 2>&1
python tools/knob_ab_probe.py g_gemm_stagger 0 1 gpurun_out/stagger_ab_audio.json audio:1,2,3,4,6,8,12 > gpurun_out/stagger_ab_audio.log 2>&1
python tools/knob_ab_probe.py g_gemm_stagger 0 1 gpurun_out/stagger_ab_text.json text:9,16,32,54,64 > gpurun_out/stagger_ab_text.log 2>&1
tail -3 gpurun_out/stagger_probe.log
