#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r3c
mkdir -p $OUT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder_batch.py tests/test_gpu_encoder.py tests/test_gpu_retrieval.py -q -k "audio or retrieval or question or fused" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -6 $OUT/tests.log
timeout 600 python tools/batch_invariance_probe.py > $OUT/batch_inv.log 2>&1
cat $OUT/batch_inv.log | tail -40
timeout 300 python tools/joint_probe.py > $OUT/joint.log 2>&1
tail -5 $OUT/joint.log
