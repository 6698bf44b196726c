#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r3f
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -8 $OUT/tests.log
timeout 900 python tools/fold_ab_probe.py $OUT/forward_ab.json > $OUT/forward_ab.log 2>&1
tail -13 $OUT/forward_ab.log
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f/bench_line.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","step_mfma_frac")}, d["roofline"]["frac"], d["scan"]["value"], d["joint_vision_audio"]["pairs_per_s"])
for k in d["kernels"]: print(k)
print(d["scan"]["retrieval"]); print(d.get("parity_vs_oracle")); print(d["cpu_baseline"])
PY
