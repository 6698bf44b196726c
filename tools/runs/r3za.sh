#!/bin/bash
mkdir -p gpurun_out/r3za
cd tools
timeout 900 python text_latency_probe.py ../gpurun_out/r3za/text_latency.json > ../gpurun_out/r3za/text_latency.log 2>&1
grep "^{" ../gpurun_out/r3za/text_latency.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3za/text_latency.log | head -30
