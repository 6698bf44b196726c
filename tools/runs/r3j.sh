#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r3j
mkdir -p $OUT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder_batch.py -q -k "attention or fused" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
for i in 1 2; do
HMM_PROBE_LIB=$REPO/tools/libhippomm_probe_noswz.so timeout 300 python tools/vswz_ab_probe.py plain >> $OUT/vswz.log 2>&1
HMM_PROBE_LIB=$REPO/tools/libhippomm_probe.so timeout 300 python tools/vswz_ab_probe.py swizzled >> $OUT/vswz.log 2>&1
done
grep "^\[" $OUT/vswz.log
