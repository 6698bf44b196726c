#!/bin/bash
# end-of-round refresh at HEAD: in-kernel stamps of the four GEMMs, plain bench line (reads the r3 PMC summaries)
mkdir -p gpurun_out/r3q
timeout 300 python tools/gemm_stamp_probe.py gpurun_out/r3q/gemm_stamps.json > gpurun_out/r3q/gemm_stamps.log 2>&1
grep -c "in_kernel_clock" gpurun_out/r3q/gemm_stamps.log
timeout 900 python bench.py > gpurun_out/r3q/bench_line.json 2> gpurun_out/r3q/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3q/bench_line.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","step_mfma_frac")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"], d["scan"]["value"], d["scan"]["roofline"]["traffic_source"], d["joint_vision_audio"]["pairs_per_s"])
s=json.load(open("gpurun_out/r3q/gemm_stamps.json"))
for k,v in s.items(): print(k, v["kernel_ms_plain"], v["fill_plus_mainloop_us"], v["in_kernel_clock_GHz_fill_plus_mainloop"], v["epilogue_us"], v["cu_handover_gap_us"])
PY
