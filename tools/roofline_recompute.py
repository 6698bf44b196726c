"""Roofline fractions straight from a `rocprofv3 --kernel-trace` CSV of bench.py (no number taken from bench.py's own line):
for the kernels DESIGN section 4 prices -- fc1 + GELU, fc2, out-proj, fused in_proj + attention at 256 frames (the full-batch
chain of bench.py's `roofline` leg and the half-batch launches of the timed step), the 1M-row scan, the bf16 prefilter pass and
the 16-query pass -- kernel, workgroups, calls, mean launch time, algorithmic FLOPs or bytes per launch, achieved rate, fraction of
the peak (2.5 PFLOP/s dense bf16, 8 TB/s).  out-proj and fc2 share a kernel symbol and a grid (N = 1280): their launches are told
apart by duration (K = 1280 vs 5120, a factor of ~3 in time).
    python3 tools/roofline_recompute.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys

PEAK_TF, PEAK_GBS = 2500.0, 8000.0
T, D, MLP, H, DH = 257, 1280, 5120, 16, 80


def main():
    acc = {}
    for r in csv.DictReader(open(sys.argv[1])):
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        acc.setdefault((r["Kernel_Name"], wg), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

    def pick(sub, wg):
        for (name, w), v in acc.items():
            if sub in name and w == wg:
                return v
        return None

    rows = []

    def add(label, sub, wg, work, unit, split=None):
        v = pick(sub, wg)
        if not v:
            return
        if split is not None:                                   # bimodal: out-proj (short) / fc2 (long) behind one symbol and grid
            cut = (min(v) * max(v)) ** 0.5
            v = [x for x in v if (x > cut) == (split == "long")]
            if not v:
                return
        avg = sum(v) / len(v)
        rate = work / avg / (1e6 if unit == "flop" else 1e3)    # TFLOP/s or GB/s
        peak = PEAK_TF if unit == "flop" else PEAK_GBS
        rows.append({"kernel": label, "symbol_contains": sub, "workgroups": wg, "calls": len(v), "avg_us": round(avg, 2),
                     "min_us": round(min(v), 2), "work_per_launch": work, "work_unit": "FLOP" if unit == "flop" else "B",
                     "achieved": round(rate, 1), "achieved_unit": "TFLOP/s" if unit == "flop" else "GB/s", "peak": peak,
                     "frac": round(rate / peak, 4)})

    for tag, frames in (("256 frames, one chain", 256), ("128 frames: half batch, BESIDE the other chain's launches", 128)):
        rows_m = frames * T
        # gemm_bf16 peels the last row tile(s) so that the main launch is whole rounds of 256 workgroups: the grid that ran says how
        # many 256-row tiles it covered
        for sub, label, tn, k_dim, n_dim, split in (("gemm_bf16_pp_kernel<1>", "fc1 + GELU", MLP // 256, D, MLP, None),
                                                    ("gemm_bf16_pp_kernel<2>", "fc2 + residual", D // 256, MLP, D, "long"),
                                                    ("gemm_bf16_pp_kernel<2>", "out-proj + residual", D // 256, D, D, "short")):
            for wg in sorted({w for (n, w) in acc if sub in n and w % tn == 0 and 0 <= (rows_m + 255) // 256 - w // tn <= 2}):
                add(f"{label} ({tag})", sub, wg, 2.0 * (wg // tn * 256) * n_dim * k_dim, "flop", split)
        add(f"fused in_proj + attention ({tag})", "qkv_attention_kernel<hmm::VisionGeo>", frames * H,
            frames * H * (2.0 * T * 3 * DH * D + 4.0 * T * T * DH), "flop")
    def full_pass_grid(sub):                                    # the grid of the symbol's 1M-row passes = the one with the most launches
        grids = {w: len(v) for (n, w), v in acc.items() if sub in n and w > 64}      # (384 / 768 workgroups since round 6, 2048 before)
        return max(grids, key=grids.get) if grids else 0

    add("scan_topk_kernel (1M x 1024 fp32 rows)", "scan_topk_kernel", full_pass_grid("scan_topk_kernel"), 1_000_000 * 4096.0, "bytes")
    add("prefilter_topk_kernel (1M x 1024 bf16 shadow rows)", "prefilter_topk_kernel", full_pass_grid("prefilter_topk_kernel"),
        1_000_000 * 2048.0, "bytes")
    add("scan_multi_kernel (1M rows once for 16 queries)", "scan_multi_kernel", 256, 1_000_000 * 4096.0, "bytes")
    add("topk_final_kernel (finish of the exact query)", "topk_final_kernel", 1, 2048 * 32 * 8.0, "bytes")
    add("prefilter_final_kernel (finish of the prefilter query)", "prefilter_final_kernel", 1, 2048 * 64 * 8.0, "bytes")
    doc = {"what": "recomputed from the kernel trace alone: mean launch duration per (kernel symbol, grid) x algorithmic work per launch "
                   "(2 M N K per GEMM on the rows the launch covers; store bytes read once per scan pass); the 128-frame rows are the timed step's launches, "
                   "which run two chains side by side: a launch there has about half the chip, its fraction is not a kernel quality figure; "
                   "finishing kernels are listed for their durations (their byte figure is the candidate lists, not a roofline claim)",
           "peaks": {"bf16_dense_TFLOPs": PEAK_TF, "hbm_GBps": PEAK_GBS}, "kernels": rows}
    print(json.dumps(doc, indent=1))
    if len(sys.argv) > 2:
        json.dump(doc, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
