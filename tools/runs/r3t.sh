#!/bin/bash
mkdir -p gpurun_out/r3t
python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_segments.py -q > gpurun_out/r3t/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3t/tests.log
python - > gpurun_out/r3t/retrieval.log 2>&1 <<'PY'
import torch, json, sys
sys.path.insert(0, ".")
import bench
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(bench.SCAN_ROWS, 1024, dtype=torch.float32, device="cuda")
for s in range(0, bench.SCAN_ROWS, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
print(json.dumps(bench.retrieval_bench(rows, True)))
PY
tail -1 gpurun_out/r3t/retrieval.log
