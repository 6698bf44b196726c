#!/bin/bash
# GPU session r6z: round 5's build against this round's, interleaved in one process: whole forwards, then the scan legs
OUT=$PWD/gpurun_out/r6z
mkdir -p $OUT
timeout 900 python tools/forward_ab_r5_probe.py $OUT/forward_ab_r5.json 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python tools/scan_ab_r5_probe.py $OUT/scan_ab_final.json 2>&1 | tail -16
