#!/bin/bash
mkdir -p gpurun_out/r3zj
cd tools
timeout 1200 python split_min_probe.py > ../gpurun_out/r3zj/split_min.log 2>&1
grep "B=" ../gpurun_out/r3zj/split_min.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zj/split_min.log | head -30
