#!/bin/bash
# check at HEAD: build() + smoke(), the whole GPU suite, the bench line
mkdir -p gpurun_out/r3zd
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r3zd/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3zd/smoke.log
python -m pytest tests -m gpu -q > gpurun_out/r3zd/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3zd/tests.log
python bench.py > gpurun_out/r3zd/bench.json 2> gpurun_out/r3zd/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3zd/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3zd/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["joint_vision_audio"]["pairs_per_s"])
print(json.dumps(d["scan"]["retrieval"])[:900])
PY
