#!/bin/bash
# GPU session r6z: hmm_rank_segment_hits -- tests, and the retrieval leg of the bench
OUT=$PWD/gpurun_out/r6z
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_scan_prefilter.py tests/test_gpu_segments.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|Error" $OUT/tests.log | tail -3
timeout 600 python - <<'PY'
import json, torch, bench
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(1_000_000, 1024, device="cuda")
for s in range(0, 1_000_000, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
r = bench.retrieval_bench(rows, True)
print(json.dumps({k: v for k, v in r.items() if k.startswith("ms_") or "equal" in k or k in ("parity_vs_oracle",)}))
PY
