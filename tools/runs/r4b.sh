#!/bin/bash
# GPU session r4b: grid-barrier / launch-floor probe; full GPU suite after the folded-LayerNorm removal
REPO=$PWD
OUT=$REPO/gpurun_out/r4b
mkdir -p $OUT
timeout 300 python tools/grid_barrier_probe.py $OUT/grid_barrier.json > $OUT/grid_barrier.log 2>&1
tail -45 $OUT/grid_barrier.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
