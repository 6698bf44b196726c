"""Audit of the few-row GEMM dispatcher: for every GEMM of a vision block at 2 ... 12 frames' rows (and audio at 2 ... 8 segments, text at
10 ... 40 questions) the dispatcher's choice against every named tile geometry, alone with COLD weights; prints the shapes where some
geometry beats the choice by more than 5 %.  Candidates only: a rule is kept after an A/B in the forwards (tools/knob_ab_probe.py).
usage: dispatch_audit_probe.py [out.json] [vision|audio|text] [few]

       dispatch_audit_probe.py --check [out.json]
The dispatcher is FROZEN (round 6).  --check reads the rules table out of the probe build (kDispatchRules in gemm_bf16.hip:
knob, shipped limit, off value, the forwards it fires on) and re-times every rule on its own forwards: wall clock per forward
with the rule at its limit and switched off, interleaved in three rounds in ONE process, embeddings compared bit for bit.  A rule
whose best gain is below 2 % on this run is to be deleted from the table (not re-tuned)."""
import json
import sys
import time

import torch

from probe_common import load_probe, event_ms, setter

L, lib = load_probe()


def check(out_path):
    import ctypes as C
    from hippomm_amd.encoder import HipTower, synthetic_state_dict
    lib.hmm_probe_dispatch_rules.restype = C.c_char_p
    rules = json.loads(lib.hmm_probe_dispatch_rules().decode())
    by_tower = {}
    for r in rules:
        for part in r["forwards"].split(";"):
            tower, batches = part.split(":")
            for b in batches.split(","):
                by_tower.setdefault(tower, []).append((r, int(b)))

    def wall_ms(fn, iters):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / iters * 1e3

    rows = []
    for kind, items in by_tower.items():
        tower = HipTower(kind, synthetic_state_dict((kind,), seed=99))
        for r, B in items:
            if kind == "text":
                x = torch.randint(1, 49000, (B, 77), device="cuda")
                x[:, 0], x[:, 20] = 49406, 49407
            elif kind == "audio":
                x = torch.randn(B, 3, 1, 128, 204, device="cuda")
            else:
                x = torch.randn(B, 3, 224, 224, device="cuda")
            out = torch.empty(B, 1024, device="cuda")
            set_knob = setter(lib, r["knob"])
            iters = 40 if B <= 32 else 8
            ms, outs = {"on": [], "off": []}, {}
            for _ in range(3):
                for tag, v in (("on", r["limit"]), ("off", r["off"])):
                    set_knob(v)
                    ms[tag].append(wall_ms(lambda: tower.forward_into(x, out), iters))
                    outs[tag] = out.clone()
            set_knob(r["limit"])
            on, off = min(ms["on"]), min(ms["off"])
            rows.append({"knob": r["knob"], "limit": r["limit"], "off": r["off"], "tower": kind, "batch": B, "ms_on": round(on, 4),
                         "ms_off": round(off, 4), "gain_pct": round((off - on) / off * 100, 2),
                         "same_bits": bool(torch.equal(outs["on"], outs["off"]))})
            print(json.dumps(rows[-1]), flush=True)
        del tower
        torch.cuda.empty_cache()
    verdict = []
    for r in rules:
        mine = [x for x in rows if x["knob"] == r["knob"]]
        best = max(x["gain_pct"] for x in mine)
        verdict.append({"knob": r["knob"], "limit": r["limit"], "what": r["what"], "evidence": r["evidence"], "best_gain_pct": best,
                        "worst_gain_pct": min(x["gain_pct"] for x in mine), "keep": best >= 2.0,
                        "bits_equal_everywhere": all(x["same_bits"] for x in mine)})
        print(json.dumps(verdict[-1]), flush=True)
    doc = {"what": "every rule of the frozen few-row dispatcher (kDispatchRules) on / off on its own forwards, interleaved, one process; "
                   "gain_pct = (off - on) / off; keep = best gain >= 2 % on this box", "rules": verdict, "forwards": rows}
    if out_path:
        json.dump(doc, open(out_path, "w"), indent=1)


if "--check" in sys.argv:
    check(next((a for a in sys.argv[1:] if not a.startswith("--")), None))
    sys.exit(0)

EPI = {"bias": 0, "gelu": 1, "resid": 2}
TILES = {"auto": -1, "auto_tiled": -2, "sliver": 5, "ring32": 8, "ring32_k2": 9, "ring32_k4": 10, "db128": 0, "pp": 3, "ring128": 6, "ring64": 7, "ring64_k2": 11, "r128x64": 12, "r64x128": 13, "r128x128w8": 14, "r128x64w8": 15}
st = L.stream_ptr()
which = sys.argv[2] if len(sys.argv) > 2 else "vision"
few = len(sys.argv) > 3 and sys.argv[3] == "few"            # the few-row end: one sample ... a handful
if which == "vision":
    D, rows_per, counts = 1280, 257, ((1, 2) if few else range(2, 13))
elif which == "audio":
    D, rows_per, counts = 768, 687, ((1,) if few else range(2, 9))
else:
    D, rows_per, counts = 1024, 77, ((1, 2, 3, 4, 6, 9) if few else (10, 12, 14, 16, 20, 24, 28, 32, 40, 48))
SHAPES = [("qkv", 3 * D, D, "bias"), ("fc1", 4 * D, D, "gelu"), ("out", D, D, "resid"), ("fc2", D, 4 * D, "resid")]
rows = []
for name, N, K, epi in SHAPES:
    copies = max(4, int(400e6 // (N * K * 2)) + 1)
    g = torch.Generator(device="cuda").manual_seed(N + K)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(copies)]
    bias = torch.randn(N, device="cuda", generator=g)
    for k in counts:
        M = k * rows_per
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        c = torch.randn(M, N, device="cuda", generator=g) if epi == "resid" else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        rec = {"tower": which, "gemm": name, "samples": k, "M": M, "N": N, "K": K}
        state = {"i": 0}
        for rep in range(2):                                   # the second pass runs at steady clocks and is the one kept
            for tag, tile in TILES.items():
                if tag == "pp" and (N % 256 or K % 128):
                    continue
                if tag in ("ring64", "ring64_k2") and ((M + 63) // 64) * (N // 64) > 1100:
                    continue
                if tag == "ring64_k2" and (K // 64) % 2:
                    continue
                if tag in ("sliver", "ring32", "ring32_k2", "ring32_k4") and (not few or ((M + 31) // 32) * (N // 32) > 2048):
                    continue
                if (tag == "ring32_k2" and (K // 64) % 2) or (tag == "ring32_k4" and (K // 64) % 4):
                    continue

                def call():
                    w = ws[state["i"] % copies]
                    state["i"] += 1
                    L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, EPI[epi], tile, st), "gemm")
                rec["us_" + tag] = round(event_ms(call, 2 * copies, warmup=copies) * 1e3, 2)
        best = min((v, t) for t, v in rec.items() if t.startswith("us_") and t != "us_auto")
        rec["best"], rec["auto_over_best"] = best[1][3:], round(rec["us_auto"] / best[0], 3)
        rows.append(rec)
        flag = "  <-- " if rec["auto_over_best"] > 1.05 else ""
        print(json.dumps(rec) + flag, flush=True)
        if len(sys.argv) > 1:
            json.dump(rows, open(sys.argv[1], "w"), indent=1)
    del ws
