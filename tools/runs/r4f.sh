#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r4f
mkdir -p $OUT
timeout 600 python tools/mid_gemm_probe.py > $OUT/mid_gemm.log 2>&1
cat $OUT/mid_gemm.log | grep "^{"
