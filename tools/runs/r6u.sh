#!/bin/bash
# GPU session r6u: the round's evidence at one commit -- full GPU suite, smoke(), bench.py under rocprofv3 + plain (profile_bench.sh:
# kernel stats, stats by grid, roofline_recompute.json), the four PMC passes, the 8-rank rehearsal, the prefilter stress, the frozen dispatcher's recheck
REPO=$PWD
OUT=$REPO/gpurun_out/r6u
mkdir -p $OUT
git rev-parse HEAD > $OUT/head.txt 2>/dev/null || true
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 bash tools/profile_bench.sh r6 > $OUT/profile_bench.log 2>&1
echo "profile_bench rc=$?"
PREFIX=r6 timeout 1500 bash tools/pmc_passes.sh r6 > $OUT/pmc.log 2>&1
echo "pmc rc=$?"; tail -3 $OUT/pmc.log
timeout 900 python tools/rehearse_n8.py $OUT/rehearsal_8_ranks.json > $OUT/rehearse.log 2>&1
echo "rehearse rc=$?"; tail -2 $OUT/rehearse.log
timeout 900 python tools/prefilter_stress.py 1000 > $OUT/stress.log 2>&1; echo "stress rc=$?"; tail -4 $OUT/stress.log
timeout 900 python tools/stress_determinism.py > $OUT/stress_determinism.log 2>&1; echo "determinism rc=$?"; tail -5 $OUT/stress_determinism.log
timeout 1500 python tools/dispatch_audit_probe.py --check $OUT/dispatch_recheck.json > $OUT/recheck.log 2>&1
echo "recheck rc=$?"; grep '"keep"' $OUT/recheck.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['knob'], d['best_gain_pct'], d['worst_gain_pct'], d['keep'])"
python3 - <<'PY'
import json
l = json.loads(open("gpurun_out/r6_bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step")}, l["roofline"]["frac"], l["roofline"].get("traffic"))
s = l["scan"]; print("scan", s["ms_per_query"], s["roofline"]["ms_per_launch"], s["roofline"]["frac"], "prefilter", s["prefilter_bf16_shadow"]["ms_per_query"], "multi", s["batched_16_queries"]["ms_per_pass"])
print("retrieval", {k: v for k, v in s["retrieval"].items() if k.startswith("ms_end") or k.startswith("device")})
f = l["formation_from_files"]; print("formation", f["paths_32"]["ms_end_to_end"], f["paths_256"]["ms_end_to_end"], f["paths_256"]["frames_per_s"], f["paths_256"]["ratio_to_tensor_in"])
a = l["audio_from_wav"]; print("audio", a["wav_1"], a["wav_16"]["ms_end_to_end"])
print(l["reference_call_sizes"]["vision_frames_32_ms"], l["reference_call_sizes"]["vision_frame_1_ms"], l["reference_call_sizes"]["audio_segment_1_ms"], l["reference_call_sizes"]["text_question_1_ms"])
PY
cd tools && timeout 300 python prefilter_final_stamps_probe.py $OUT/prefilter_final_stamps.json > /dev/null; cd ..
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $REPO/tools/scan_trace_workload.py > $OUT/scan_traced.json 2> $OUT/scan_traced.err
cd $REPO
python3 tools/scan_trace_summarize.py "$(find $OUT/prof -name '*kernel_trace.csv' | head -1)" $OUT/scan_trace_summary.json | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items(): print(k, v['period_us_median'], v['kernels_us_median'])"
rm -rf $OUT/prof
