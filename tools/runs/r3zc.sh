#!/bin/bash
mkdir -p gpurun_out/r3zc
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_encoder_batch.py tests/test_gpu_retrieval.py -x -q > gpurun_out/r3zc/tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r3zc/tests.log
cd tools
timeout 900 python small_gemm_ab_probe.py ../gpurun_out/r3zc/small_gemm_ab.json > ../gpurun_out/r3zc/small_gemm_ab.log 2>&1
grep "B=" ../gpurun_out/r3zc/small_gemm_ab.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zc/small_gemm_ab.log | head -30
