#!/bin/bash
# GPU session r5f: the round's last check at HEAD -- full GPU suite, smoke(), the default bench line (no profiler)
REPO=$PWD
OUT=$REPO/gpurun_out/r5f
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
l = json.loads(open("gpurun_out/r5f/bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus")}, l["roofline"]["frac"], l.get("reference_call_sizes", {}))
PY
