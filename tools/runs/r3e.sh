#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r3e
mkdir -p $OUT
A="--workload cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-scan"
HMM_BENCH_DUMP=/tmp/n1a.npy python bench.py $A --gpus 1 > $OUT/n1a.json 2> $OUT/n1a.err
HMM_BENCH_DUMP=/tmp/n1b.npy python bench.py $A --gpus 1 > $OUT/n1b.json 2> $OUT/n1b.err
HMM_BENCH_REHEARSAL=1 HMM_BENCH_DUMP=/tmp/n2a.npy python bench.py $A --gpus 2 > $OUT/n2a.json 2> $OUT/n2a.err
HMM_BENCH_REHEARSAL=1 HMM_BENCH_DUMP=/tmp/n2b.npy python bench.py $A --gpus 2 > $OUT/n2b.json 2> $OUT/n2b.err
HMM_BENCH_REHEARSAL=1 HMM_BENCH_DUMP=/tmp/n4.npy python bench.py $A --gpus 4 > $OUT/n4.json 2> $OUT/n4.err
python - <<'PY' > $OUT/compare.log 2>&1
import numpy as np
m = {k: np.load(f"/tmp/{k}.npy") for k in ("n1a", "n1b", "n2a", "n2b", "n4")}
for a, b in (("n1a", "n1b"), ("n2a", "n2b"), ("n1a", "n2a"), ("n1a", "n4"), ("n2a", "n4")):
    d = (m[a] != m[b]).any(axis=1)
    rows = np.nonzero(d)[0]
    print(a, b, "rows differing:", rows.size, "first", rows[:20].tolist(), "last", rows[-5:].tolist() if rows.size else [])
    if rows.size:
        print("   max abs diff", float(np.abs(m[a] - m[b]).max()), "elements differing in first bad row", int((m[a][rows[0]] != m[b][rows[0]]).sum()))
PY
cat $OUT/compare.log
timeout 900 python tools/split_probe.py > $OUT/split.log 2>&1
tail -12 $OUT/split.log
