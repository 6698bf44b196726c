#!/bin/bash
mkdir -p gpurun_out/r3zh
cd tools
timeout 900 python fused_small_probe.py > ../gpurun_out/r3zh/fused_small.log 2>&1
grep "B=" ../gpurun_out/r3zh/fused_small.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zh/fused_small.log | head -30
