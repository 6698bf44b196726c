#!/bin/bash
mkdir -p gpurun_out/r3zb
cd tools
timeout 900 python small_gemm_ab_probe.py ../gpurun_out/r3zb/small_gemm_ab.json > ../gpurun_out/r3zb/small_gemm_ab.log 2>&1
grep "B=" ../gpurun_out/r3zb/small_gemm_ab.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zb/small_gemm_ab.log | head -30
