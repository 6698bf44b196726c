#!/bin/bash
mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_scan_prefilter.py -x -q > gpurun_out/r4w/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4w/tests.log
timeout 600 python tools/prefilter_probe.py gpurun_out/r4w/prefilter_probe.json 2>&1 | grep "^{"
timeout 1500 python tools/prefilter_stress.py 300 > gpurun_out/r4w/prefilter_stress.log 2>&1; echo "prefilter rc=$?"; tail -5 gpurun_out/r4w/prefilter_stress.log
