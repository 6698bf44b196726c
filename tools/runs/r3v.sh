#!/bin/bash
mkdir -p gpurun_out/r3v
python -m pytest tests/test_gpu_scan.py tests/test_gpu_live_golden.py -q > gpurun_out/r3v/tests.log 2>&1; echo "tests rc=$?"; grep -v "amdgpu.ids" gpurun_out/r3v/tests.log | tail -4
