#!/bin/bash
# GPU session r6w: micro-settings of the pipeline's main loop, interleaved twice
OUT=$PWD/gpurun_out/r6w
mkdir -p $OUT
timeout 1200 python tools/formation_probe.py $OUT/formation_probe.json > $OUT/probe.log 2>&1
echo "probe rc=$?"; grep '^{"frames"' $OUT/probe.log | cut -c1-330
