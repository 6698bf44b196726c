#!/bin/bash
mkdir -p gpurun_out/r3s
timeout 300 python tools/quant_probe.py > gpurun_out/r3s/quant.log 2>&1; grep -v amdgpu gpurun_out/r3s/quant.log | tail -5
