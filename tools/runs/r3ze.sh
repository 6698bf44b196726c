#!/bin/bash
mkdir -p gpurun_out/r3ze
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k gemm > gpurun_out/r3ze/tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r3ze/tests.log
cd tools
timeout 600 python sliver_probe.py ../gpurun_out/r3ze/sliver.json > ../gpurun_out/r3ze/sliver.log 2>&1
python - <<'PY'
import json
for r in json.load(open("../gpurun_out/r3ze/sliver.json")):
    print(r["gemm"], r["M"], "128:", r["us_128x128"], "ring128:", r["us_128x128_ring4"], "ring64:", r["us_64x64_ring4"], r["ring64_same_bits"], "ring32:", r["us_32x32_ring4"], r["ring32_same_bits"], "sliver:", min(r["us_sliver_16"], r["us_sliver_32"], r["us_sliver_64"]), "auto:", r["us_auto"])
PY
grep -i "error\|Traceback" -A 8 ../gpurun_out/r3ze/sliver.log | head -30
#timeout 900 python text_latency_probe.py ../gpurun_out/r3ze/text_latency.json > ../gpurun_out/r3ze/text_latency.log 2>&1
grep "^{" ../gpurun_out/r3ze/text_latency.log; grep -i "error\|Traceback" -A 8 ../gpurun_out/r3ze/text_latency.log | head -30
