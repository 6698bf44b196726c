#!/bin/bash
# GPU session r6j: pipeline tests again + the probe after the interpreter-lock diet (whole-file read, one decode call, quota-aware workers)
OUT=$PWD/gpurun_out/r6j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_preprocess.py tests/test_gpu_formation.py tests/test_audio_fbank.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 1200 python tools/formation_probe.py $OUT/formation_probe.json > $OUT/probe.log 2>&1
echo "probe rc=$?"; tail -60 $OUT/probe.log
