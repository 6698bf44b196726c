#!/bin/bash
# GPU session r6v: the quick route of the prefilter's finish (k-th block maximum as the threshold when it yields few candidates)
REPO=$PWD
OUT=$REPO/gpurun_out/r6v
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_scan_prefilter.py tests/test_gpu_segments.py tests/test_gpu_retrieval.py tests/test_gpu_live_golden.py tests/test_gpu_scan.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 600 python tools/prefilter_stress.py 300 > $OUT/stress.log 2>&1; echo "stress rc=$?"; tail -4 $OUT/stress.log
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
