#!/bin/bash
# audio one chain at 5 vs 6 segments: kernel timelines (why is 5 the only size where one chain beats two?)
REPO=$PWD
OUT=$REPO/gpurun_out/r5v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "audio 5 one" "audio 6 one" "audio 5" ; do
  set -- $spec
  tag=$1_$2_${3:-two}
  rm -rf $OUT/tr_$tag
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -- python3 $REPO/tools/trace_forward.py $1 $2 $3 > $OUT/tr_$tag.log 2>&1
  f=$(find $OUT/tr_$tag -name '*kernel_trace.csv' | head -1)
  python3 $REPO/tools/timeline.py $f im2col_audio $OUT/timeline_$tag.json > $OUT/timeline_$tag.txt 2>&1
  rm -rf $OUT/tr_$tag
  head -14 $OUT/timeline_$tag.txt | cut -c1-160
done
