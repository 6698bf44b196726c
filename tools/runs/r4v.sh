#!/bin/bash
# GPU session r4v: race / soak runs for the round's new kernels (profiles/r4_ring_stress.log, r4_prefilter_stress.log)
mkdir -p gpurun_out/r4v
timeout 1500 python tools/prefilter_stress.py 300 > gpurun_out/r4v/prefilter_stress.log 2>&1; echo "prefilter rc=$?"; tail -5 gpurun_out/r4v/prefilter_stress.log
timeout 2400 python tools/ring_stress.py 1500 > gpurun_out/r4v/ring_stress.log 2>&1; echo "ring rc=$?"; tail -4 gpurun_out/r4v/ring_stress.log
