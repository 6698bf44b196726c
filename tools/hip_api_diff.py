"""Per-step HIP API calls of bench.py's timed region from two `rocprofv3 --hip-trace --stats` runs that differ only in
--steps.  usage: hip_api_diff.py <dir steps=a> <dir steps=b> <b - a> <out json>"""
import csv, glob, json, os, sys
a_dir, b_dir, dsteps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def calls(d):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*hip_api_stats.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Name"]] = acc.get(r["Name"], 0) + int(r["Calls"])
    return acc


a, b = calls(a_dir), calls(b_dir)
per_step = {k: (b.get(k, 0) - a.get(k, 0)) / dsteps for k in sorted(set(a) | set(b)) if b.get(k, 0) != a.get(k, 0)}
sync_like = {k: v for k, v in per_step.items() if any(t in k for t in ("Malloc", "Free", "Memcpy", "Synchronize", "EventQuery", "StreamQuery"))}
doc = {"command": "rocprofv3 --hip-trace --stats --output-format csv -- python3 bench.py --steps {2,12} --warmup 2 --no-cpu-baseline --no-scan",
       "what": "HIP API calls per timed step = (calls at --steps 12 - calls at --steps 2) / 10; everything outside the timed loop "
               "is identical in both runs and cancels",
       "calls_per_step": per_step, "allocation_copy_or_sync_calls_per_step": sync_like,
       "launch_only": all(abs(v) < 0.15 for v in sync_like.values())}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
