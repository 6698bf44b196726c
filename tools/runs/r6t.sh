#!/bin/bash
# GPU session r6t (host only): one burst of 32 frames on T threads / P forked processes: where does a burst's time go?
OUT=$PWD/gpurun_out/r6t
mkdir -p $OUT
timeout 600 python tools/decode_burst_probe.py $OUT/decode_burst.json 2>&1 | tail -16
