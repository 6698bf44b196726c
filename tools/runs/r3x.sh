#!/bin/bash
mkdir -p gpurun_out/r3x
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > gpurun_out/r3x/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r3x/tests.log
cd tools && timeout 600 python sliver_probe.py ../gpurun_out/r3x/sliver.json > ../gpurun_out/r3x/sliver.log 2>&1; cd ..
grep "^{" gpurun_out/r3x/sliver.log; grep -i "error\|Traceback" -A 5 gpurun_out/r3x/sliver.log | head -20
