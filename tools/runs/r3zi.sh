#!/bin/bash
mkdir -p gpurun_out/r3zi
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_encoder_batch.py tests/test_gpu_retrieval.py tests/test_gpu_configs.py -x -q > gpurun_out/r3zi/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed" gpurun_out/r3zi/tests.log | tail -2
cd tools
timeout 900 python text_latency_probe.py ../gpurun_out/r3zi/text_latency.json > ../gpurun_out/r3zi/text_latency.log 2>&1
grep "^{" ../gpurun_out/r3zi/text_latency.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zi/text_latency.log | head -30
