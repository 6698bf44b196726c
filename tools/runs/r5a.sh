#!/bin/bash
# GPU session r5a: smoke, the profiled + plain bench lines, the PMC passes, the 8-rank rehearsal -- all at HEAD
REPO=$PWD
OUT=$REPO/gpurun_out/r5a
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/profile_bench.sh r5final > $OUT/profile.log 2>&1
tail -1 $OUT/profile.log | cut -c1-200
PREFIX=r5 bash tools/pmc_passes.sh r5 > $OUT/pmc.log 2>&1
tail -3 $OUT/pmc.log
timeout 1500 python tools/rehearse_n8.py $OUT/rehearsal_8_ranks.json > $OUT/rehearse.log 2>&1; echo "rehearsal rc=$?"
