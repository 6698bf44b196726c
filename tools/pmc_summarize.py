"""Turn the rocprofv3 --pmc CSVs of tools/pmc_workload.py into the summaries committed under profiles/.
usage: pmc_summarize.py <dir with mfma/ fetch/ write/ sub-directories> <out prefix, e.g. profiles/r2>
  mfma/  : --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  fetch/ : --pmc FETCH_SIZE        write/ : --pmc WRITE_SIZE   (separate passes: TCC has 4 slots, FETCH_SIZE takes 3)
gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts the 128-B requests of wide coalesced reads as 64 B
-> x2; both counters are in KB and count fabric requests, Infinity-Cache hits included."""
import csv, glob, json, os, sys
from collections import defaultdict

src, prefix = sys.argv[1], sys.argv[2]


def load(sub):
    """kernel key -> counter -> values per dispatch.  The two 1280-workgroup residual GEMMs (out-proj K=1280, fc2 K=5120) share
    a kernel name and a grid: the workload launches them alternately, so they are told apart by dispatch parity."""
    disp = defaultdict(dict)                                # (name, grid) -> dispatch id -> {counter: value, "_ns": duration}
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            d = disp[(r["Kernel_Name"], int(r["Grid_Size"]))].setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = defaultdict(lambda: defaultdict(list))
    for (name, grid), ds in disp.items():
        split = "gemm_bf16_pp_kernel<2>" in name and grid == 1280 * 512
        # the prefilter path enqueues the exact scan's two kernels behind a device-side flag: they return at once (a few us) and
        # must not be averaged with the real launches
        cond = "scan_topk_kernel" in name or "topk_final_kernel" in name
        for i, did in enumerate(sorted(ds)):
            skipped = cond and ds[did].get("_ns", 1e9) < (50e3 if "scan_topk_kernel" in name else 6e3)
            key = (name + (" [out_proj]" if i % 2 == 0 else " [fc2]") if split else name + (" [conditional: skipped]" if skipped else ""), grid)
            for c, v in ds[did].items():
                rows[key][c].append(v)
    return rows


def short(name):
    if "[conditional: skipped]" in name:
        return None
    for key, tag in (("gemm_bf16_pp_kernel", "gemm_pp"), ("gemm_bf16_kernel", "gemm_small"), ("qkv_attention_kernel", "qkv_attention"),
                     ("attention_kernel", "attention"), ("layernorm", "layernorm"), ("scan_topk_kernel", "scan_topk"),
                     ("scan_multi_kernel", "scan_multi"), ("topk_final", "topk_final"), ("prefilter_topk_kernel", "prefilter_topk"),
                     ("prefilter_final_kernel", "prefilter_final")):
        if key in name:
            return tag
    return None


mean = lambda v: sum(v) / len(v)
mfma, fetch, write = load("mfma"), load("fetch"), load("write")
# --- MFMA utilisation
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- "
                  "python3 tools/pmc_workload.py",
       "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): share of SIMD-cycles with the "
                     "matrix pipe busy while the GPU was active (GRBM_GUI_ACTIVE is summed over the 8 XCDs; the counter adds 16 "
                     "cycles per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16 / f32 16x16x4).  clock_GHz = GRBM_GUI_ACTIVE / 8 / kernel "
                     "time: the board runs these kernels at its power cap and the clock gives way, so mfma_busy_over_peak_at_2p4GHz (the fraction of the 2.5 PF/s "
                     "dense peak the matrix pipes were kept busy for) = mfma_util x clock / 2.4.  Profiled passes run a few % slower "
                     "than unprofiled ones.",
       "kernels": []}
for (name, grid), c in sorted(mfma.items()):
    tag = short(name)
    if tag is None or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    busy, gui, ns = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(c["GRBM_GUI_ACTIVE"]), mean(c["_ns"])
    out["kernels"].append({"kernel": tag, "full_name": name[:140], "grid_threads": grid, "dispatches": len(c["GRBM_GUI_ACTIVE"]),
                           "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui,
                           "SQ_BUSY_CYCLES": mean(c["SQ_BUSY_CYCLES"]) if "SQ_BUSY_CYCLES" in c else None,
                           "duration_us_profiled": round(ns / 1e3, 1),
                           "clock_GHz": round(gui / 8 / ns, 3) if ns else None,     # GRBM_GUI_ACTIVE / 8 / kernel time
                           "mfma_util": round(busy / (gui / 8 * 256 * 4), 4) if gui else None,
                           "mfma_busy_over_peak_at_2p4GHz": round(busy / (ns * 2.4 * 256 * 4), 4) if ns else None})
json.dump(out, open(prefix + "_mfma_util_pmc_summary.json", "w"), indent=1)
# --- traffic
traffic = []
for key in sorted(set(fetch) | set(write)):
    tag = short(key[0])
    if tag is None:
        continue
    f = mean(fetch[key]["FETCH_SIZE"]) if key in fetch and "FETCH_SIZE" in fetch[key] else None
    w = mean(write[key]["WRITE_SIZE"]) if key in write and "WRITE_SIZE" in write[key] else None
    traffic.append({"kernel": tag, "full_name": key[0][:120], "grid_threads": key[1], "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                    "bytes_per_launch": (2 * f * 1024 if f else 0) + (w * 1024 if w else 0)})
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 tools/pmc_workload.py "
                      "(separate passes)",
           "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads; KB units; fabric requests, Infinity-Cache hits included",
           "kernels": traffic}, open(prefix + "_traffic_pmc_all.json", "w"), indent=1)
# --- LDS bank conflicts (optional fourth pass: lds/ with --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE)
lds = load("lds")
if lds:
    rows_out = []
    for (name, grid), c in sorted(lds.items()):
        tag = short(name)
        if tag is None or "SQ_LDS_IDX_ACTIVE" not in c:
            continue
        conf, act = mean(c.get("SQ_LDS_BANK_CONFLICT", [0.0])), mean(c["SQ_LDS_IDX_ACTIVE"])
        rows_out.append({"kernel": tag, "full_name": name[:120], "grid_threads": grid, "SQ_LDS_BANK_CONFLICT": conf,
                         "SQ_LDS_IDX_ACTIVE": act, "conflict_share_of_lds_cycles": round(conf / act, 4) if act else None})
    json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -- python3 "
                          "tools/pmc_workload.py",
               "definition": "SQ_LDS_BANK_CONFLICT = extra LDS cycles spent on bank conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array "
                             "cycles (MI355X_MICROARCH.md); their ratio per kernel",
               "kernels": rows_out}, open(prefix + "_lds_pmc_summary.json", "w"), indent=1)
# the two files bench.py reads (kernel name + traffic_bytes_per_launch)
def pick(tag, grid_min=0):
    c = [t for t in traffic if t["kernel"] == tag and t["grid_threads"] >= grid_min]
    return max(c, key=lambda t: t["grid_threads"]) if c else None
fc1_main = [t for t in traffic if t["kernel"] == "gemm_pp" and t["grid_threads"] == 5120 * 512]
for t in traffic:
    if "_ns" in t:
        del t["_ns"]
fc1_tail = [t for t in traffic if t["kernel"] == "gemm_small" and t["grid_threads"] == 80 * 256]
if fc1_main:
    total = fc1_main[0]["bytes_per_launch"] + (fc1_tail[0]["bytes_per_launch"] if fc1_tail else 0)
    json.dump({"kernel": "gemm_bf16[mlp_fc1+gelu]", "traffic_bytes_per_launch": total, "parts": fc1_main + fc1_tail[:1],
               "algorithmic_bytes_per_launch": 2 * (65792 * 1280 + 5120 * 1280 + 65792 * 5120),
               "note": "M=65792, N=5120, K=1280: ping-pong kernel on 256 row tiles (5120 workgroups) + peeled tail (80 workgroups); "
                       "fabric-side requests, Infinity-Cache hits included"},
              open(prefix + "_gemm_pmc_summary.json", "w"), indent=1)
sc = pick("scan_topk")
if sc:
    json.dump({"kernel": "scan_topk_kernel", "traffic_bytes_per_launch": sc["bytes_per_launch"], "parts": [sc],
               "algorithmic_bytes_per_launch": 4096000000}, open(prefix + "_scan_pmc_summary.json", "w"), indent=1)
pf = pick("prefilter_topk")
if pf:
    json.dump({"kernel": "prefilter_topk_kernel", "traffic_bytes_per_launch": pf["bytes_per_launch"], "parts": [pf],
               "algorithmic_bytes_per_launch": 2048000000}, open(prefix + "_prefilter_pmc_summary.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
print(json.dumps(traffic, indent=1))
