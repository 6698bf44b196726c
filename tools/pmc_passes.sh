#!/bin/bash
# The PMC passes behind profiles/*_pmc_*: one rocprofv3 run per counter group (TCC has 4 slots, FETCH_SIZE takes 3), counters only
# with --kernel-trace.  Run on the GPU box from the repo root:  bash tools/pmc_passes.sh <tag>  -> gpurun_out/pmc_<tag>/{mfma,fetch,write,lds}
set -e
TAG=${1:-final}
REPO=$PWD
OUT="$REPO/gpurun_out/pmc_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 "$REPO/tools/pmc_workload.py" > "$OUT/mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$REPO/tools/pmc_workload.py" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$REPO/tools/pmc_workload.py" > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/lds" -- python3 "$REPO/tools/pmc_workload.py" > "$OUT/lds.log" 2>&1
cd "$REPO"
mkdir -p "gpurun_out/pmc_${TAG}_summary"
python3 tools/pmc_summarize.py "$OUT" "gpurun_out/pmc_${TAG}_summary/${PREFIX:-r3}"
ls "gpurun_out/pmc_${TAG}_summary"
# keep only what the summariser read (the merge back is capped at 64 MiB)
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
du -sh "$OUT"
