#!/bin/bash
# GPU session r6x: smoke() with the files-in leg, and soaks at the final kernels
OUT=$PWD/gpurun_out/r6x
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python tools/prefilter_stress.py 1000 > $OUT/prefilter_stress.log 2>&1; echo "prefilter stress rc=$?"; tail -4 $OUT/prefilter_stress.log
timeout 900 python tools/stress_determinism.py > $OUT/stress_determinism.log 2>&1; echo "determinism rc=$?"; tail -5 $OUT/stress_determinism.log
