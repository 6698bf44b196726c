#!/bin/bash
# GPU session r6z: the streaming kernels at their final settings against round 5's build (interleaved, one process), then the scan tests
OUT=$PWD/gpurun_out/r6z
mkdir -p $OUT
timeout 900 python tools/scan_ab_r5_probe.py $OUT/scan_ab_final.json 2>&1 | tail -16
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_scan_prefilter.py tests/test_gpu_live_golden.py tests/test_gpu_segments.py tests/test_gpu_retrieval.py tests/test_gpu_multi_query.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
