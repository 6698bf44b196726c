#!/bin/bash
# GPU session r5t: kernel timelines of the reference's call sizes at the round's final kernels (1 question, 1 frame, 1 segment, 12 / 20 / 32 frames)
REPO=$PWD
OUT=$REPO/gpurun_out/r5t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "text 1 embed_tokens" "vision 1 im2col_vision" "audio 1 im2col_audio" "vision 12 im2col_vision" "vision 20 im2col_vision" "vision 32 im2col_vision"; do
  set -- $spec
  rm -rf $OUT/tr_$1_$2
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$1_$2 -- python3 $REPO/tools/trace_forward.py $1 $2 > $OUT/tr_$1_$2.log 2>&1
  f=$(find $OUT/tr_$1_$2 -name '*kernel_trace.csv' | head -1)
  python3 $REPO/tools/timeline.py $f $3 $OUT/timeline_$1_$2.json > $OUT/timeline_$1_$2.txt 2>&1
  rm -rf $OUT/tr_$1_$2
done
