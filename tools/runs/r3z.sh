#!/bin/bash
mkdir -p gpurun_out/r3z
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_encoder_batch.py -x -q > gpurun_out/r3z/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r3z/tests.log
cd tools
timeout 600 python sliver_probe.py ../gpurun_out/r3z/sliver.json > ../gpurun_out/r3z/sliver.log 2>&1
grep "^{" ../gpurun_out/r3z/sliver.log | cut -c1-400; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3z/sliver.log | head -30
timeout 900 python text_latency_probe.py ../gpurun_out/r3z/text_latency.json > ../gpurun_out/r3z/text_latency.log 2>&1
grep "^{" ../gpurun_out/r3z/text_latency.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3z/text_latency.log | head -30
