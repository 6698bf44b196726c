#!/bin/bash
# HIP runtime environment settings against the few-row forwards (launch-boundary bound): one process per setting, each under
# its own timeout -- ROC_SYSTEM_SCOPE_SIGNAL=0 never came back in round 5 (the call ran into gpurun's limit) and is not in the list
mkdir -p gpurun_out
out=gpurun_out/env_knobs.jsonl
: > $out
for rep in 1; do
  timeout 120 python tools/call_sizes_probe.py default >> $out 2>/dev/null
  HIP_FORCE_DEV_KERNARG=1 timeout 120 python tools/call_sizes_probe.py dev_kernarg_1 >> $out 2>/dev/null
  HIP_FORCE_DEV_KERNARG=0 timeout 120 python tools/call_sizes_probe.py dev_kernarg_0 >> $out 2>/dev/null
  AMD_OPT_FLUSH=0 timeout 120 python tools/call_sizes_probe.py opt_flush_0 >> $out 2>/dev/null
  DEBUG_HIP_KERNARG_COPY_OPT=0 timeout 120 python tools/call_sizes_probe.py kernarg_copy_opt_0 >> $out 2>/dev/null
  DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 timeout 120 python tools/call_sizes_probe.py hdp_flush_wa_0 >> $out 2>/dev/null
  ROC_USE_FGS_KERNARG=0 timeout 120 python tools/call_sizes_probe.py fgs_kernarg_0 >> $out 2>/dev/null
done
python - <<'PY'
import json
for l in open("gpurun_out/env_knobs.jsonl"):
    r = json.loads(l)
    print(r["label"], {k: v for k, v in r.items() if k.endswith("_ms")})
PY
