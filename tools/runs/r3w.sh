#!/bin/bash
mkdir -p gpurun_out/r3w
timeout 400 python tools/tile_shape_probe.py gpurun_out/r3w/tile_shape.json > gpurun_out/r3w/tile_shape.log 2>&1
grep "^{" gpurun_out/r3w/tile_shape.log; grep -i "error\|Traceback" gpurun_out/r3w/tile_shape.log | head -3
