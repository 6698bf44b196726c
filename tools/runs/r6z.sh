#!/bin/bash
# GPU session r6z: the new grids of the streaming kernels (384 / 768 workgroups): every scan test, the stress, against round 5's build
OUT=$PWD/gpurun_out/r6z
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_scan_prefilter.py tests/test_gpu_live_golden.py tests/test_gpu_segments.py tests/test_gpu_retrieval.py tests/test_gpu_multi_query.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 900 python tools/prefilter_stress.py 1000 > $OUT/stress.log 2>&1; echo "stress rc=$?"; tail -4 $OUT/stress.log
timeout 900 python tools/scan_ab_r5_probe.py $OUT/scan_ab_r5.json 2>&1 | tail -15
