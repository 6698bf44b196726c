#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r3d
mkdir -p $OUT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_folded_layernorm.py tests/test_gpu_encoder_batch.py -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -6 $OUT/tests.log
timeout 300 python tools/fold_ln_kernel_probe.py > $OUT/fold_kernels.log 2>&1
tail -13 $OUT/fold_kernels.log
timeout 600 python tools/cfg5_invariance_probe.py > $OUT/cfg5_inv.log 2>&1
tail -12 $OUT/cfg5_inv.log
timeout 900 python tools/fold_ab_probe.py $OUT/fold_ab.json > $OUT/fold_ab.log 2>&1
tail -16 $OUT/fold_ab.log
