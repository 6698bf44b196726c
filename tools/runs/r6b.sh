#!/bin/bash
# GPU session r6b: host-only probe of the decode scaling on the GPU box's host (cgroup quota? interpreter lock? allocator?)
OUT=$PWD/gpurun_out/r6b
mkdir -p $OUT
timeout 900 python tools/decode_scaling_probe.py $OUT/decode_scaling.json > $OUT/probe.log 2>&1
echo "rc=$?"; cat $OUT/probe.log
