#!/bin/bash
mkdir -p gpurun_out/r3zg
cd tools
timeout 900 python tail_ab_probe.py > ../gpurun_out/r3zg/tail_ab.log 2>&1
grep "B=" ../gpurun_out/r3zg/tail_ab.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zg/tail_ab.log | head -30
