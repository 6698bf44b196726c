"""FETCH_SIZE per tile walk from `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/walk_evidence_probe.py --pmc`.
usage: walk_pmc_summarize.py <rocprof out dir> <out json>.  The workload launches, per shape (fc1, fc2, qkv) and per walk in
WALKS order, 6 launches of the ping-pong GEMM; the last 4 of each group are averaged.  gfx950: FETCH_SIZE is in KB and counts
128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, sys
src, out = sys.argv[1], sys.argv[2]
rows = []
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_bf16_pp_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]),
                         (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3))
rows.sort()
WALKS = ["strips", "8x4", "4x8", "16x2", "6x5", "3x10"]
shapes = [("fc1", 5120, 1280, 2), ("fc2", 1280, 5120, 8), ("qkv", 3840, 1280, 2)]
M = 65536
res, i = {}, 0
for name, N, K, out_b in shapes:
    algo = 2 * (M * K + N * K)
    for tag in WALKS:
        c = int(tag.split("x")[1]) if "x" in tag else 0
        if N // 256 < c:
            continue
        grp = rows[i:i + 6]; i += 6
        if len(grp) < 6:
            break
        kb = sum(g[2] for g in grp[2:]) / 4
        res[f"{name}_{tag}"] = {"FETCH_SIZE_KB": round(kb, 1), "fetched_bytes_x2": int(2 * kb * 1024), "operand_bytes_algorithmic": algo,
                                "fetch_over_operands": round(2 * kb * 1024 / algo, 2), "kernel_us_profiled": round(sum(g[3] for g in grp[2:]) / 4, 1)}
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 tools/walk_evidence_probe.py --pmc",
           "M": M, "walks": res, "dispatches_seen": len(rows)}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
