#!/bin/bash
mkdir -p gpurun_out/r3zm
cd tools
timeout 1700 python ring_stress.py 3000 > ../gpurun_out/r3zm/ring_stress.log 2>&1; echo "rc=$?"
tail -16 ../gpurun_out/r3zm/ring_stress.log
