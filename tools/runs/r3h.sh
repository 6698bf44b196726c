#!/bin/bash
# final profiles of the round at HEAD: bench under rocprofv3 --kernel-trace --stats + plain bench line; PMC passes
REPO=$PWD
bash tools/profile_bench.sh r3 > gpurun_out/r3_profile_bench.log 2>&1
tail -3 gpurun_out/r3_profile_bench.log
PREFIX=r3 bash tools/pmc_passes.sh r3 > gpurun_out/r3_pmc_passes.log 2>&1
tail -5 gpurun_out/r3_pmc_passes.log
ls gpurun_out/pmc_r3_summary
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_line.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","step_mfma_frac")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["scan"]["value"], d["scan"]["roofline"]["frac"], d["joint_vision_audio"]["pairs_per_s"])
PY
