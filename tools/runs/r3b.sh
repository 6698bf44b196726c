#!/bin/bash
# GPU session r3b: spill-free RESID_XB epilogue; fold / LN-nt / walk A/B in the forward; new tests (configs, retrieval, live golden)
REPO=$PWD
OUT=$REPO/gpurun_out/r3b
mkdir -p $OUT
python -m pytest tests/test_gpu_folded_layernorm.py tests/test_gpu_configs.py tests/test_gpu_retrieval.py tests/test_gpu_live_golden.py tests/test_gpu_select.py tests/test_gpu_ops.py -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -8 $OUT/tests.log
timeout 300 python tools/fold_ln_kernel_probe.py > $OUT/fold_kernels.log 2>&1
tail -13 $OUT/fold_kernels.log
timeout 900 python tools/fold_ab_probe.py $OUT/fold_ab.json > $OUT/fold_ab.log 2>&1
tail -16 $OUT/fold_ab.log
