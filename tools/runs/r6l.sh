#!/bin/bash
# GPU session r6l: (a) the frozen dispatcher's recheck, (b) kernel trace of back-to-back scan queries: where are the 34-44 us?
REPO=$PWD
OUT=$REPO/gpurun_out/r6l
mkdir -p $OUT
timeout 1500 python tools/dispatch_audit_probe.py --check $OUT/dispatch_recheck.json > $OUT/recheck.log 2>&1
echo "recheck rc=$?"; grep '"keep"' $OUT/recheck.log
timeout 300 python tools/scan_trace_workload.py > $OUT/scan_plain.json 2>$OUT/scan_plain.err; cat $OUT/scan_plain.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $REPO/tools/scan_trace_workload.py > $OUT/scan_traced.json 2> $OUT/scan_traced.err
echo "trace rc=$?"; cat $OUT/scan_traced.json
cd $REPO
python3 tools/scan_trace_summarize.py "$(find $OUT/prof -name '*kernel_trace.csv' | head -1)" $OUT/scan_trace_summary.json
rm -rf $OUT/prof
