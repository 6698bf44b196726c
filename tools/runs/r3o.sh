#!/bin/bash
mkdir -p gpurun_out/r3o
timeout 200 python tools/power_probe.py > gpurun_out/r3o/power_probe.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3o/power_probe.log | tail -22
timeout 120 python tools/dma_power_probe.py gpurun_out/r3o/dma_power.json > gpurun_out/r3o/dma_power.log 2>&1
grep "^{" gpurun_out/r3o/dma_power.log
