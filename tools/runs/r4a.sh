#!/bin/bash
# GPU session r4a: kernel timelines of the reference's call sizes (1 question, 1 frame, 1 segment, 32 frames)
REPO=$PWD
OUT=$REPO/gpurun_out/r4a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "text 1 embed_tokens" "vision 1 im2col_vision" "audio 1 im2col_audio" "vision 32 im2col_vision"; do
  set -- $spec
  rm -rf $OUT/tr_$1_$2
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$1_$2 -- python3 $REPO/tools/trace_forward.py $1 $2 > $OUT/tr_$1_$2.log 2>&1
  f=$(find $OUT/tr_$1_$2 -name '*kernel_trace.csv' | head -1)
  python3 $REPO/tools/timeline.py $f $3 $OUT/timeline_$1_$2.json > $OUT/timeline_$1_$2.txt 2>&1
  head -30 $OUT/timeline_$1_$2.txt
  rm -rf $OUT/tr_$1_$2
done
