#!/bin/bash
# Is a timed bench.py step launch-only?  Two rocprofv3 --hip-trace --stats runs of the same command that differ only in
# --steps (2 and 12): the per-API call-count difference / 10 is what ONE step issues.  Run on the GPU box from the repo
# root:  bash tools/hip_api_passes.sh <tag>  -> gpurun_out/<tag>_hip_api_per_step.json
set -e
TAG=${1:-final}
REPO=$PWD
OUT="$REPO/gpurun_out/hipapi_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for K in 2 12; do
  rocprofv3 --hip-trace --stats --output-format csv -d "$OUT/steps$K" -- python3 "$REPO/bench.py" --steps $K --warmup 2 --no-cpu-baseline --no-scan \
      > "$OUT/steps$K.json" 2> "$OUT/steps$K.err"
done
cd "$REPO"
python3 tools/hip_api_diff.py "$OUT/steps2" "$OUT/steps12" 10 "gpurun_out/${TAG}_hip_api_per_step.json"
find "$OUT" -name "*.csv" -size +5M -delete; find "$OUT" -name "*.db" -delete
