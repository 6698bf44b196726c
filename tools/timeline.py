"""Timeline of the last forward in a `rocprofv3 --kernel-trace` CSV: per kernel launch its start (us after the forward's
first kernel), duration and the gap since the previous kernel ended (any stream).  python3 tools/timeline.py <kernel_trace.csv>
<first-kernel-substring> [out.json]  -- the forward is delimited by launches whose name contains the substring."""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("hmm::", "").replace("void ", "")
    return name[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
    # forwards start at a marker that is not directly preceded (within 20 launches) by another marker of the same forward
    starts = [m for j, m in enumerate(marks) if j == 0 or m - marks[j - 1] > 20]
    if len(starts) < 2:
        raise SystemExit("fewer than two forwards in the trace")
    lo, hi = starts[-2], starts[-1]              # the second to last forward, complete
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = t0
    out = []
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        out.append({"kernel": short(r["Kernel_Name"]), "wgs": wg, "start_us": round((s - t0) / 1e3, 2),
                    "dur_us": round((e - s) / 1e3, 2), "gap_us": round((s - prev_end) / 1e3, 2), "stream": r.get("Stream_Id", "")})
        prev_end = max(prev_end, e)
    total = (prev_end - t0) / 1e3
    busy = sum(o["dur_us"] for o in out)
    gaps = sum(max(0.0, o["gap_us"]) for o in out)
    print(f"launches {len(out)}  span {total:.1f} us  sum of durations {busy:.1f} us  sum of positive gaps {gaps:.1f} us")
    agg = {}
    for o in out:
        a = agg.setdefault(o["kernel"], [0, 0.0, 0.0])
        a[0] += 1; a[1] += o["dur_us"]; a[2] += max(0.0, o["gap_us"])
    for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:4d} x {d / n:8.2f} us  (+ gap before {g / n:6.2f})  {k}")
    if len(sys.argv) > 3:
        json.dump({"span_us": total, "busy_us": busy, "gap_us": gaps, "launches": out}, open(sys.argv[3], "w"), indent=0)


if __name__ == "__main__":
    main()
