#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r3g
mkdir -p $OUT
timeout 600 python tools/priority_probe.py > $OUT/priority.log 2>&1
tail -8 $OUT/priority.log
timeout 600 python tools/layer_probe.py fused=1 streams=2,1 --json $OUT/layer.json > $OUT/layer.log 2>&1
tail -4 $OUT/layer.log
bash tools/hip_api_passes.sh r3g > $OUT/hipapi.log 2>&1
tail -30 $OUT/hipapi.log
