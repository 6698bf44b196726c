#!/bin/bash
# GPU session r3a: folded-LayerNorm parity + cfg tests, fold / tail A/B, walk evidence (time, stamps, FETCH_SIZE), bench line
REPO=$PWD
OUT=$REPO/gpurun_out/r3a
mkdir -p $OUT
python -m pytest tests/test_gpu_folded_layernorm.py tests/test_gpu_configs.py tests/test_gpu_encoder_batch.py tests/test_gpu_encoder.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 600 python tools/fold_ab_probe.py $OUT/fold_ab.json > $OUT/fold_ab.log 2>&1
tail -22 $OUT/fold_ab.log
timeout 300 python tools/fold_ln_kernel_probe.py > $OUT/fold_kernels.log 2>&1
tail -12 $OUT/fold_kernels.log
timeout 400 python tools/walk_evidence_probe.py $OUT/walk_evidence.json > $OUT/walk_evidence.log 2>&1
tail -16 $OUT/walk_evidence.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/walk_pmc -- python3 $REPO/tools/walk_evidence_probe.py --pmc > $OUT/walk_pmc.log 2>&1
cd $REPO
python tools/walk_pmc_summarize.py $OUT/walk_pmc $OUT/walk_fetch.json > $OUT/walk_fetch.log 2>&1
tail -30 $OUT/walk_fetch.log
find $OUT/walk_pmc -name "*.csv" -size +20M -delete; find $OUT/walk_pmc -name "*.db" -delete
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3a/bench_line.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","step_mfma_frac")}, d["roofline"]["frac"], d["scan"]["value"], d["joint_vision_audio"]["pairs_per_s"])
for k in d["kernels"]: print(k)
PY
