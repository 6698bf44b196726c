#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r4p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder_batch.py tests/test_gpu_encoder.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 600 python tools/text_latency_probe.py $OUT/text_latency.json > $OUT/text_latency.log 2>&1
python - <<'PY'
import json
for d in json.load(open("gpurun_out/r4p/text_latency.json")):
    print(d["tower"], d["batch"], d.get("ms_eager_round3_product"), d.get("ms_eager_deepk"), d.get("ms_eager_product"), d.get("same_bits"))
PY
