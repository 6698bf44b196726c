#!/bin/bash
# GPU session r4o: full GPU suite + smoke at HEAD, then the profiled + plain bench lines
REPO=$PWD
OUT=$REPO/gpurun_out/r4o
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/profile_bench.sh r4final > $OUT/profile.log 2>&1
tail -1 $OUT/profile.log | cut -c1-200
