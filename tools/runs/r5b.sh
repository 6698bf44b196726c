#!/bin/bash
# GPU session r5b: full GPU suite + smoke at HEAD, then everything of r5a (profiled + plain bench lines, PMC passes, 8-rank rehearsal)
REPO=$PWD
OUT=$REPO/gpurun_out/r5b
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
bash tools/runs/r5a.sh
