#!/bin/bash
# GPU session r6k: full GPU suite + smoke + the default bench line at the pipeline commit
OUT=$PWD/gpurun_out/r6k
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r6k/bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step")}, l["roofline"]["frac"])
print(json.dumps(l.get("formation_from_files"))[:3000])
print(json.dumps(l.get("audio_from_wav"))[:1500])
print(json.dumps(l["scan"]["retrieval"])[:600])
PY
