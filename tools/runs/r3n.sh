#!/bin/bash
mkdir -p gpurun_out/r3n
timeout 120 python tools/dma_power_probe.py gpurun_out/r3n/dma_power.json > gpurun_out/r3n/dma_power.log 2>&1
grep "^{" gpurun_out/r3n/dma_power.log; tail -3 gpurun_out/r3n/dma_power.log | grep -v "^{"
