#!/bin/bash
mkdir -p gpurun_out/r3u
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_bench_rehearsal.py -q > gpurun_out/r3u/tests.log 2>&1; echo "tests rc=$?"; grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r3u/tests.log | tail -12
ls /dev/dri 2>&1 | head; ls /sys/class/kfd/kfd/topology/nodes 2>&1 | head -20; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo $n $(grep -E "simd_count|drm_render_minor" $n/properties | tr '\n' ' '); done 2>&1 | head -20
