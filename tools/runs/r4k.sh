#!/bin/bash
# GPU session r4k: full GPU suite (prefilter re-score batching, query split, store cache); latency probe; bench line
REPO=$PWD
OUT=$REPO/gpurun_out/r4k
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -8 $OUT/tests.log
timeout 600 python tools/text_latency_probe.py $OUT/text_latency.json > $OUT/text_latency.log 2>&1
python - <<'PY'
import json
for d in json.load(open("gpurun_out/r4k/text_latency.json")):
    print(d["tower"], d["batch"], d.get("ms_eager_round3_product"), d.get("ms_eager_deepk"), d.get("ms_eager_product"), d.get("same_bits"))
PY
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4k/bench_line.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","step_mfma_frac","step_mfma_frac_executed")}, d["roofline"]["frac"], d["scan"]["value"], d["scan"]["prefilter_bf16_shadow"]["ms_per_query"], d["scan"]["batched_16_queries"]["hbm_frac"])
c=d["reference_call_sizes"]; print({k:v for k,v in c.items() if k.endswith("_ms")})
print(d["scan"]["retrieval"])
PY
