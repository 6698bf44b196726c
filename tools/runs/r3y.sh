#!/bin/bash
mkdir -p gpurun_out/r3y
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_encoder_batch.py -x -q > gpurun_out/r3y/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r3y/tests.log
cd tools && timeout 600 python text_latency_probe.py ../gpurun_out/r3y/text_latency.json > ../gpurun_out/r3y/text_latency.log 2>&1; cd ..
grep "^{" gpurun_out/r3y/text_latency.log; grep -i "error\|Traceback" -A 8 gpurun_out/r3y/text_latency.log | head -30
