#!/bin/bash
mkdir -p gpurun_out/r3p
timeout 200 python tools/dma_power_probe.py gpurun_out/r3p/dma_power.json > gpurun_out/r3p/dma_power.log 2>&1
grep "^{" gpurun_out/r3p/dma_power.log; grep -i "error\|Traceback" gpurun_out/r3p/dma_power.log | head -3
timeout 400 python tools/mfma_power_probe.py gpurun_out/r3p/mfma_power.json > gpurun_out/r3p/mfma_power.log 2>&1
grep "^{" gpurun_out/r3p/mfma_power.log; grep -i "error\|Traceback" gpurun_out/r3p/mfma_power.log | head -3
