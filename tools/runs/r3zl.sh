#!/bin/bash
mkdir -p gpurun_out/r3zl
cd tools
timeout 1200 python mid_batch_probe.py > ../gpurun_out/r3zl/mid_batch.log 2>&1
grep "B=" ../gpurun_out/r3zl/mid_batch.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zl/mid_batch.log | head -30
