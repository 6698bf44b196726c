#!/bin/bash
# GPU session r4l: smoke(), the scan / prefilter / ops tests touched last, profiled + plain bench lines at HEAD
REPO=$PWD
OUT=$REPO/gpurun_out/r4l
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_scan_prefilter.py tests/test_gpu_ops.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
bash tools/profile_bench.sh r4final > $OUT/profile.log 2>&1
tail -2 $OUT/profile.log | cut -c1-300
