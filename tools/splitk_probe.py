"""Round 5: the few-row forwards (one frame, one question, one audio segment) with (a) the column-major tile list of the ring
kernels (every weight byte leaves HBM / the Infinity Cache once instead of once per row tile) and (b) deterministic split-K for
out-proj / fc2 with the reduction inside the LayerNorm behind them.

Part A: single GEMMs with COLD weights (a new weight copy per call, enough copies to exceed the 256-MiB Infinity Cache).
Part B: whole forwards, wall clock, every knob combination, cosine of the embeddings against the non-split build.
usage: splitk_probe.py [out.json] [--part a|b|ab]"""
import json
import sys
import time

import torch

from probe_common import load_probe, setter, event_ms

L, lib = load_probe()
set_col = setter(lib, "g_gemm_col_major")
set_rows = setter(lib, "g_enc_splitk_rows")
set_rows_v = setter(lib, "g_enc_splitk_rows_vision")
_set_fc2_va = setter(lib, "g_enc_splitk_fc2")
_set_fc2_t = setter(lib, "g_enc_splitk_fc2_text")


def set_fc2(v):
    _set_fc2_va(v)
    _set_fc2_t(v)
set_out = setter(lib, "g_enc_splitk_out")
set_sktile = setter(lib, "g_gemm_splitk_tile")
set_rect = setter(lib, "g_gemm_rect")
EPI = {"bias": 0, "gelu": 1, "resid": 2}
TILE = {"auto": -1, "ring128": 6, "ring64": 7, "ring32": 8, "ring32_k2": 9, "ring32_k4": 10, "ring64_k2": 11}
out_path = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
part_sel = "ab"
if "--part" in sys.argv:
    part_sel = sys.argv[sys.argv.index("--part") + 1]
rows_out = []


def emit(rec):
    rows_out.append(rec)
    print(json.dumps(rec), flush=True)
    if out_path:
        json.dump(rows_out, open(out_path, "w"), indent=1)


def part_a():
    st = L.stream_ptr()
    shapes = [("text qkv", 3072, 1024, "bias"), ("text out", 1024, 1024, "resid"), ("text fc1", 4096, 1024, "gelu"),
              ("text fc2", 1024, 4096, "resid"), ("vision qkv", 3840, 1280, "bias"), ("vision out", 1280, 1280, "resid"),
              ("vision fc1", 5120, 1280, "gelu"), ("vision fc2", 1280, 5120, "resid"), ("audio qkv", 2304, 768, "bias"),
              ("audio out", 768, 768, "resid"), ("audio fc1", 3072, 768, "gelu"), ("audio fc2", 768, 3072, "resid")]
    for name, N, K, epi in shapes:
        copies = max(4, int(400e6 // (N * K * 2)) + 1)
        Ms = (77, 154, 308) if name.startswith("text") else ((687, 1374) if name.startswith("audio") else (257, 514, 1028))
        for M in Ms:
            g = torch.Generator(device="cuda").manual_seed(M + N + K)
            a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(copies)]
            bias = torch.randn(N, device="cuda", generator=g)
            x0 = torch.randn(M, N, device="cuda", generator=g)
            gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
            rec = {"gemm": name, "M": M, "N": N, "K": K}
            state = {"i": 0}
            c = x0.clone() if epi == "resid" else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")

            def plain(tile=-1, with_ln=False):
                def call():
                    w = ws[state["i"] % copies]
                    state["i"] += 1
                    L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, EPI[epi], tile, st), "gemm")
                    if with_ln:
                        L.check(lib.hmm_op_layernorm_bf16(c.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), M, N, 1e-6, st), "ln")
                return call
            for order in (0, 1):
                set_col(order)
                rec[f"us_auto_col{order}"] = round(event_ms(plain(), 3 * copies, warmup=copies) * 1e3, 2)
                if epi != "resid":
                    for tname in ("ring32", "ring64", "ring128"):
                        if tname == "ring32" and ((M + 31) // 32) * (N // 32) > 2048:
                            continue
                        rec[f"us_{tname}_col{order}"] = round(event_ms(plain(TILE[tname]), 3 * copies, warmup=copies) * 1e3, 2)
            set_col(0)
            if epi == "resid" and N in (768, 1024, 1280):
                rec["us_gemm_ln"] = round(event_ms(plain(with_ln=True), 3 * copies, warmup=copies) * 1e3, 2)
                # reference result of the pair for the error check
                c.copy_(x0)
                state["i"] = 0
                plain(with_ln=True)()
                torch.cuda.synchronize()
                ref_x, ref_y = c.clone(), y.float().clone()
                for S in (2, 4, 8):
                    if K % (64 * S):
                        continue
                    part = torch.empty(S, M, N, device="cuda")
                    for tname in ("auto", "ring32", "ring32_k2", "ring64", "ring64_k2", "ring128"):
                        kl = K // S
                        if tname.endswith("_k2") and kl % 128:
                            continue
                        if tname.startswith("ring32") and ((M + 31) // 32) * (N // 32) * S > 4096:
                            continue

                        def split():
                            w = ws[state["i"] % copies]
                            state["i"] += 1
                            L.check(lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), M, N, K, S, TILE[tname], st), "splitk")
                            L.check(lib.hmm_op_layernorm_reduce_bf16(c.data_ptr(), part.data_ptr(), S, bias.data_ptr(), gam.data_ptr(),
                                                                     bet.data_ptr(), y.data_ptr(), M, N, 1e-6, st), "lnr")
                        rec[f"us_split{S}_{tname}_ln"] = round(event_ms(split, 3 * copies, warmup=copies) * 1e3, 2)

                        def gemm_only():
                            w = ws[state["i"] % copies]
                            state["i"] += 1
                            L.check(lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), M, N, K, S, TILE[tname], st), "splitk")
                        rec[f"us_split{S}_{tname}"] = round(event_ms(gemm_only, 3 * copies, warmup=copies) * 1e3, 2)
                    c.copy_(x0)
                    state["i"] = 0
                    split()
                    torch.cuda.synchronize()
                    rec[f"split{S}_max_abs_dx"] = float((c - ref_x).abs().max())
                    rec[f"split{S}_max_abs_dy"] = float((y.float() - ref_y).abs().max())
            emit(rec)
            del ws


def wall_ms(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def part_b():
    from hippomm_amd.encoder import HipTower, synthetic_state_dict
    for kind, batches in (("text", (1, 4)), ("vision", (1, 2, 3)), ("audio", (1, 2))):
        sd = synthetic_state_dict((kind,), seed=99)
        tower = HipTower(kind, sd)
        del sd
        for B in batches:
            if kind == "text":
                x = torch.randint(1, 49000, (B, 77), device="cuda")
                x[:, 0], x[:, 20] = 49406, 49407
            elif kind == "audio":
                x = torch.randn(B, 3, 1, 128, 204, device="cuda")
            else:
                x = torch.randn(B, 3, 224, 224, device="cuda")
            rec = {"tower": kind, "batch": B}
            base = None
            variants = [("r4", 0, 0, 2, 1), ("product", 0, 700, -1, 1), ("product_no_rect", 0, 700, -2, 1), ("r4_again", 0, 0, 2, 1),
                        ("product_again", 0, 700, -1, 1), ("product_no_rect_again", 0, 700, -2, 1)]
            for tag, col, rows, fc2, outp in variants:
                set_col(col)
                set_rows(rows)
                set_rows_v(300 if rows else 0)
                set_rect(0 if fc2 == -2 or tag.startswith("r4") else 1)
                if fc2 > 0:
                    set_fc2(fc2)
                else:                                   # the product's per-tower split factors
                    _set_fc2_va(2)
                    _set_fc2_t(4)
                set_out(outp)
                out = torch.empty(B, 1024, device="cuda")
                try:
                    rec[f"ms_{tag}"] = round(wall_ms(lambda: tower.forward_into(x, out)), 4)
                except Exception as ex:     # e.g. a split that does not divide K
                    rec[f"ms_{tag}"] = str(ex)[:80]
                    continue
                if base is None:
                    base = out.clone()
                else:
                    o, b = out.double(), base.double()
                    cos = (o * b).sum(1) / (o.norm(dim=1) * b.norm(dim=1))
                    rec[f"one_minus_cos_{tag}"] = float((1 - cos).max())
                    rec[f"same_bits_{tag}"] = bool(torch.equal(out, base))
            set_col(0)
            emit(rec)
        del tower
        torch.cuda.empty_cache()


if "a" in part_sel:
    part_a()
if "b" in part_sel:
    part_b()
