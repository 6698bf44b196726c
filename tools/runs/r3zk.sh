#!/bin/bash
REPO=$PWD
mkdir -p gpurun_out/r3zk
cd /tmp && export TMPDIR=/tmp
for cfg in "vision 32" "text 1"; do
  set -- $cfg
  rm -rf /tmp/prof_$1
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$1 -- python3 $REPO/tools/trace_forward.py $1 $2 > /dev/null 2>&1
  python3 - $1 $2 <<'PY'
import csv, sys, collections, glob
kind, B = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/prof_{kind}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for r in rows:
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    name = r["Kernel_Name"].split("(")[0][-70:]
    acc[(name, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"== {kind} B={B}: per forward (6 forwards traced)")
tot = 0
for (k, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{sum(v)/6:9.1f} us/fwd  {len(v)/6:6.1f} calls/fwd  avg {sum(v)/len(v):7.2f} us  wg {wg:6d}  {k}")
    tot += sum(v) / 6
print(f"sum of listed kernels per forward: {tot:.0f} us")
# gaps: last forward only
n = len(rows) // 6
last = rows[-n:]
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last)
print(f"last forward: span {span:.0f} us, kernel time {busy:.0f} us, launches {n}")
PY
done > $REPO/gpurun_out/r3zk/trace.log 2>&1
cat $REPO/gpurun_out/r3zk/trace.log
