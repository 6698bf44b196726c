#!/bin/bash
# GPU session r6a: the files -> embeddings pipeline for the first time on the box -- its tests, then the probe (decode scaling with
# threads, the whole call per (workers, min_chunk, max_inflight), the audio call)
REPO=$PWD
OUT=$REPO/gpurun_out/r6a
mkdir -p $OUT
nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" 
timeout 900 python -m pytest tests/test_preprocess.py tests/test_gpu_formation.py tests/test_audio_fbank.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 1200 python tools/formation_probe.py $OUT/formation_probe.json --audio > $OUT/probe.log 2>&1
echo "probe rc=$?"; tail -60 $OUT/probe.log
