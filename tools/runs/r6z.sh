#!/bin/bash
# GPU session r6z: transposed butterfly in the exact scan's streaming kernel: tests (bits = the re-score's), then against round 5's build
OUT=$PWD/gpurun_out/r6z
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_scan_prefilter.py tests/test_gpu_live_golden.py tests/test_gpu_retrieval.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 900 python tools/scan_ab_r5_probe.py $OUT/scan_ab_transposed.json 2>&1 | tail -16
