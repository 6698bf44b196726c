#!/bin/bash
# GPU session r4c: graph replay vs eager launches at the reference's call sizes; GPU suite with the tightened tolerances and the
# 8-rank rehearsals; a bench line on this box
REPO=$PWD
OUT=$REPO/gpurun_out/r4c
mkdir -p $OUT
timeout 600 python tools/graph_latency_probe.py $OUT/graph_latency.json > $OUT/graph_latency.log 2>&1
tail -14 $OUT/graph_latency.log
timeout 1800 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -25 $OUT/tests.log
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c/bench_line.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","step_mfma_frac","step_mfma_frac_executed","ms_per_step_with_count_readback","startup_s")}, d["roofline"]["frac"], d["scan"]["value"], d["joint_vision_audio"]["pairs_per_s"])
print(d["reference_call_sizes"]); print(d["parity_vs_oracle"]); print(d["cpu_baseline"])
for k in d["kernels"]: print(k)
PY
