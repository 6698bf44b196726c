#!/bin/bash
# The two runs behind profiles/*_bench_*: bench.py under rocprofv3 --kernel-trace --stats (short), then plain (default).
# Run on the GPU box from the repo root:  bash tools/profile_bench.sh <tag>   -> gpurun_out/<tag>_* (kernel stats, stats by grid,
# <tag>_roofline_recompute.json = the fractions of DESIGN section 4 from the trace alone: tools/roofline_recompute.py)
set -e
TAG=${1:-final}
REPO=$PWD
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "$REPO/gpurun_out/prof_$TAG"
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_$TAG" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline \
    > "$REPO/gpurun_out/${TAG}_profiled_line.json" 2> "$REPO/gpurun_out/${TAG}_profiled.err"
cd "$REPO"
cp "$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)" "gpurun_out/${TAG}_kernel_stats.csv"
cp "$(find gpurun_out/prof_$TAG -name '*kernel_trace.csv' | head -1)" "gpurun_out/${TAG}_kernel_trace.csv"
python3 - "$TAG" <<'PY'
import csv, sys, collections
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/{tag}_kernel_trace.csv")))
acc = collections.defaultdict(list)
for r in rows:
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    acc[(r["Kernel_Name"], wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"gpurun_out/{tag}_kernel_stats_by_grid.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "workgroups", "calls", "total_us", "avg_us", "min_us", "max_us"])
    for (k, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, wg, len(v), round(sum(v), 1), round(sum(v) / len(v), 2), round(min(v), 2), round(max(v), 2)])
PY
python3 tools/roofline_recompute.py "gpurun_out/${TAG}_kernel_trace.csv" "gpurun_out/${TAG}_roofline_recompute.json" > /dev/null
rm -rf "gpurun_out/prof_$TAG" "gpurun_out/${TAG}_kernel_trace.csv"
python3 bench.py > "gpurun_out/${TAG}_bench_line.json" 2> "gpurun_out/${TAG}_bench.err"
tail -c 400 "gpurun_out/${TAG}_bench_line.json"
