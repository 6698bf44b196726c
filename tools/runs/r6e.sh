#!/bin/bash
# GPU session r6e: do ranges of one call overlap on the GPU when issued through two handles on two streams?
OUT=$PWD/gpurun_out/r6e
mkdir -p $OUT
timeout 600 python tools/concurrent_forward_probe.py $OUT/concurrent_forward.json 2>&1 | tail -40
