"""Where does a feature_search query spend its time?  From a `rocprofv3 --kernel-trace` CSV of tools/scan_trace_workload.py: for the
last 50 queries of each kind the begin / end stamps of its kernels -> per kernel the duration, and the gaps between them
(previous kernel's END to next kernel's BEGIN: what a launch boundary costs on the device's clock).
    python3 tools/scan_trace_summarize.py <kernel_trace.csv> [out.json]"""
import csv
import json
import re
import statistics
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("hmm::", "").replace("void ", "")[:60]


def med(xs):
    return round(statistics.median(xs), 2) if xs else None


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3,
           int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
    out = {}
    for tag, first in (("exact", "scan_topk_kernel"), ("prefilter", "prefilter_topk_kernel"), ("multi16", "scan_multi_kernel"),
                       ("scan_kernel_only", "scan_topk_kernel")):
        starts = [i for i, k in enumerate(ks) if k[0].startswith(first) and k[3] > 64 and k[2] - k[1] > 50]
        if tag == "exact":                                      # streaming launches that ARE followed by the finishing kernel
            starts = [i for i in starts if i + 1 < len(ks) and ks[i + 1][0].startswith("topk_final_kernel")]
        if tag == "scan_kernel_only":                           # ... and those followed by another streaming launch
            starts = [i for i in starts if i + 1 < len(ks) and ks[i + 1][0].startswith("scan_topk_kernel")
                      and (i == 0 or ks[i - 1][0].startswith("scan_topk_kernel"))]
        if len(starts) < 52:
            continue
        starts = starts[-51:]
        per_kernel, gaps, period = {}, {}, []
        for a, b in zip(starts, starts[1:]):
            period.append(ks[b][1] - ks[a][1])
            prev_end = None
            for j in range(a, b):
                name, s, e, wg = ks[j]
                per_kernel.setdefault(f"{name} [{wg} wg]", []).append(e - s)
                if prev_end is not None:
                    gaps.setdefault(f"before {name}", []).append(s - prev_end)
                prev_end = e
            gaps.setdefault(f"before next {ks[b][0]}", []).append(ks[b][1] - prev_end)
        out[tag] = {"queries": len(period), "period_us_median": med(period), "period_us_min": round(min(period), 2),
                    "kernels_us_median": {k: med(v) for k, v in per_kernel.items()},
                    "kernels_us_min": {k: round(min(v), 2) for k, v in per_kernel.items()},
                    "gaps_us_median": {k: med(v) for k, v in gaps.items()},
                    "sum_kernels_us": round(sum(statistics.median(v) for v in per_kernel.values()), 2),
                    "sum_gaps_us": round(sum(statistics.median(v) for v in gaps.values()), 2)}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
