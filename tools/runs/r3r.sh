#!/bin/bash
mkdir -p gpurun_out/r3r
timeout 400 python tools/stress_determinism.py 3 > gpurun_out/r3r/stress.log 2>&1; echo "stress rc=$?"; grep -v amdgpu.ids gpurun_out/r3r/stress.log | tail -6
