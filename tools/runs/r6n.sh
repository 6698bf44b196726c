#!/bin/bash
# GPU session r6n: phases of prefilter_final_kernel (stamps) + trace with the scan-kernel-only group
REPO=$PWD
OUT=$REPO/gpurun_out/r6n
mkdir -p $OUT
cd tools && timeout 300 python prefilter_final_stamps_probe.py $OUT/prefilter_final_stamps.json; cd ..
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $REPO/tools/scan_trace_workload.py > $OUT/scan_traced.json 2> $OUT/scan_traced.err
echo "trace rc=$?"; cat $OUT/scan_traced.json
cd $REPO
python3 tools/scan_trace_summarize.py "$(find $OUT/prof -name '*kernel_trace.csv' | head -1)" $OUT/scan_trace_summary.json | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items(): print(k, v['period_us_median'], v['period_us_min'], v['kernels_us_median'])"
rm -rf $OUT/prof
