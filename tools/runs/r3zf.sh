#!/bin/bash
mkdir -p gpurun_out/r3zf
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_encoder_batch.py tests/test_gpu_retrieval.py -x -q > gpurun_out/r3zf/tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r3zf/tests.log
cd tools
timeout 600 python sliver_probe.py ../gpurun_out/r3zf/sliver.json > ../gpurun_out/r3zf/sliver.log 2>&1
python - <<'PY'
import json
for r in json.load(open("../gpurun_out/r3zf/sliver.json")):
    best = min(r["us_128x128_ring4"], r["us_64x64_ring4"], r["us_32x32_ring4"], r["us_sliver_16"], r["us_sliver_32"], r["us_sliver_64"], r["us_128x128"])
    print(r["gemm"], r["M"], "ring128:", r["us_128x128_ring4"], "ring64:", r["us_64x64_ring4"], "ring32:", r["us_32x32_ring4"], "auto:", r["us_auto"], "best:", best, "" if r["us_auto"] <= best * 1.08 + 0.3 else "  <-- auto misses")
PY
grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zf/sliver.log | head -30
timeout 900 python text_latency_probe.py ../gpurun_out/r3zf/text_latency.json > ../gpurun_out/r3zf/text_latency.log 2>&1
grep "^{" ../gpurun_out/r3zf/text_latency.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zf/text_latency.log | head -30
timeout 900 python small_gemm_ab_probe.py ../gpurun_out/r3zf/small_gemm_ab.json > ../gpurun_out/r3zf/small_gemm_ab.log 2>&1
grep "B=" ../gpurun_out/r3zf/small_gemm_ab.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3zf/small_gemm_ab.log | head -30
