#!/usr/bin/env python3
"""Benchmark of the HippoMM hot path on MI355X (contract: see the round brief).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): ImageBind-huge ViT-H/14 vision tower, bf16 MFMA, 256 synthetic
224x224 frames per GPU per step.  A step = one pass of the path over one batch: encode the local 256
frames -> (N > 1: one RCCL all-gather of the (256,1024) fp32 embeddings) -> cosine key-frame
selection on the gathered matrix.  Weak scaling: per-GPU work is fixed, value = N*256*K / time.

The same JSON line also carries the second half of BASELINE's metric under "scan": the feature_search
scan over a resident 1M x 1024 fp32 store (k=32) in GB/s of algorithmic bytes (4096 B per row).

roofline      : the dominant kernel of the step (the bf16 GEMM instance with the largest time share),
                algorithmic FLOPs per launch / mean launch time measured live with HIP events on the
                launch stream; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).
cpu_baseline  : the fp32 oracle of the same tower (oracle/imagebind_oracle.py, torch CPU) timed on this
                host's cores on a bounded sample, rank 0, N=1 only.  A reported baseline, not a target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X
PEAK_HBM_GBS = 8000.0          # HBM3E spec
FRAMES_PER_GPU = 256
SCAN_ROWS, SCAN_K = 1_000_000, 32


def event_time_ms(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_roofline(rows):
    """Time each GEMM instance of one transformer block at the step's row count; return the
    roofline object of the dominant one plus the per-kernel table."""
    import ctypes as C
    from hippomm_amd import _lib as L
    lib = L.load()
    shapes = [("qkv_proj", 3840, 1280, 0), ("out_proj+residual", 1280, 1280, 2),
              ("mlp_fc1+gelu", 5120, 1280, 1), ("mlp_fc2+residual", 1280, 5120, 2)]
    table = []
    for name, N, K, epi in shapes:
        a = torch.randn(rows, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        bias = torch.zeros(N, device="cuda")
        c = torch.zeros(rows, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)

        def run():
            L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(),
                                         rows, N, K, epi, L.stream_ptr()), "gemm")
        ms = event_time_ms(run, 10)
        flops = 2.0 * rows * N * K
        table.append({"kernel": f"gemm_bf16[{name}]", "M": rows, "N": N, "K": K, "ms": round(ms, 4),
                      "tflops": round(flops / ms / 1e9, 1), "launches_per_step": 32})
        del a, w, c
    dom = max(table, key=lambda r: r["ms"])
    roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(dom["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": None,
            "flops_per_launch": 2.0 * dom["M"] * dom["N"] * dom["K"], "ms_per_launch": dom["ms"]}
    if dom["kernel"] == "gemm_bf16[mlp_fc1+gelu]" and rows == 65792:
        # PMC passes (profiles/r1_gemm_pmc_summary.json): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, main + peeled
        # launch.  These are fabric-side requests, Infinity-Cache hits included: 4.9x the operand bytes because the strip
        # tile walk streams the 13-MB weight matrix once per 256-row tile (cache traffic, the kernel is MFMA-bound).
        roof["traffic"] = 4156223882
        roof["traffic_source"] = "profiles/r1_gemm_pmc_summary.json"
        roof["algorithmic_bytes_per_launch"] = 2 * (rows * 1280 + 5120 * 1280 + rows * 5120)
    return roof, table


def scan_bench(do_cpu):
    from hippomm_amd.vector_ops import FeatureStore
    g = torch.Generator(device="cuda").manual_seed(42)
    rows = torch.empty(SCAN_ROWS, 1024, dtype=torch.float32, device="cuda")
    for s in range(0, SCAN_ROWS, 125_000):
        blk = torch.randn(125_000, 1024, generator=g, device="cuda")
        rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
    q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
    store = FeatureStore(rows)
    ms_query = event_time_ms(lambda: store.search_device(q, SCAN_K), 20, warmup=3)
    # the streaming kernel alone (dominant kernel of the scan)
    import ctypes as C
    from hippomm_amd import _lib as L
    lib = L.load()
    cand = torch.empty(2048 * SCAN_K, dtype=torch.int64, device="cuda")
    ms_kernel = event_time_ms(lambda: L.check(lib.hmm_op_scan_topk_only(rows.data_ptr(), SCAN_ROWS, q.data_ptr(), SCAN_K,
                                                                         cand.data_ptr(), L.stream_ptr()), "scan"), 20)
    algo_bytes = SCAN_ROWS * 4096.0
    out = {
        "metric": "cosine-scan GB/s (feature_search, 1M x 1024 fp32 store, top-32, 1 query)",
        "value": round(algo_bytes / ms_query / 1e6, 1), "unit": "GB/s", "ms_per_query": round(ms_query, 4),
        "dtype": "f32",
        "roofline": {"bound": "hbm", "kernel": "scan_topk_kernel", "achieved": round(algo_bytes / ms_kernel / 1e6, 1),
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(algo_bytes / ms_kernel / 1e6 / PEAK_HBM_GBS, 4),
                     # PMC pass (profiles/r1_scan_pmc_summary.json): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE
                     "traffic": 4096643072, "traffic_source": "profiles/r1_scan_pmc_summary.json",
                     "bytes_per_launch": algo_bytes, "ms_per_launch": round(ms_kernel, 4)},
    }
    # batched questions: 16 queries per pass over the same store
    q16 = torch.randn(16, 1024, generator=torch.Generator(device="cuda").manual_seed(44), device="cuda")
    ms_multi = event_time_ms(lambda: store.search_multi_device(q16, SCAN_K), 10, warmup=3)
    out["batched_16_queries"] = {"ms_per_pass": round(ms_multi, 4), "us_per_query": round(ms_multi / 16 * 1e3, 1),
                                 "store_read_GBps": round(algo_bytes / ms_multi / 1e6, 1),
                                 "speedup_vs_16_single_scans": round(16 * ms_query / ms_multi, 2)}
    # reference point: the same query through stock PyTorch-ROCm operators on the resident store (vector_ops.py:151-188
    # moved to the GPU as is: norms, matrix-vector product, division, top-k)
    def torch_scan():
        sims = (rows @ q) / (rows.norm(dim=1) * q.norm())
        return torch.topk(sims, SCAN_K)
    ms_torch = event_time_ms(torch_scan, 10, warmup=3)
    out["torch_rocm_reference"] = {"ms_per_query": round(ms_torch, 4), "GBps": round(algo_bytes / ms_torch / 1e6, 1),
                                   "what": "rows @ q / (rows.norm(dim=1) * q.norm()) + torch.topk on the same GPU"}
    if do_cpu:
        from oracle.vector_ops_oracle import top_k_cosine_similarity_oracle
        n_cpu = 200_000
        sub = rows[:n_cpu].cpu().numpy()
        qh = q.cpu().numpy()
        top_k_cosine_similarity_oracle(qh, sub, SCAN_K)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            top_k_cosine_similarity_oracle(qh, sub, SCAN_K)
            best = min(best, time.perf_counter() - t0)
        out["cpu_baseline"] = {"value": round(n_cpu * 4096.0 / best / 1e9, 2), "unit": "GB/s",
                               "cores": os.cpu_count(), "kind": "port",
                               "sample": f"numpy oracle (vector_ops.py:151-188 restated), first {n_cpu} rows of the "
                                         f"same store, k=32, best of 3 after a warm run; numpy/BLAS default threads"}
    del rows, store
    torch.cuda.empty_cache()
    return out


def joint_bench():
    """BASELINE configs[2]: vision + audio joint encode, 128 frame / 10 s-spectrogram pairs on one GPU."""
    from hippomm_amd.encoder import HipTower, synthetic_state_dict
    pairs = 128
    sd = synthetic_state_dict(("vision", "audio"), seed=1234)
    vis, aud = HipTower("vision", sd), HipTower("audio", sd)
    del sd
    frames = torch.randn(pairs, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    mels = torch.randn(pairs, 3, 1, 128, 204, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    ev, ea = torch.empty(pairs, 1024, device="cuda"), torch.empty(pairs, 1024, device="cuda")

    def run():
        vis.forward_into(frames, ev)
        aud.forward_into(mels, ea)
    ms = event_time_ms(run, 5, warmup=2)
    flops = vis.flops(pairs) + aud.flops(pairs)
    out = {"workload": "BASELINE cfg3: 128 frame + 10 s log-mel (3 clips) pairs, vision then audio tower",
           "pairs_per_s": round(pairs / ms * 1e3, 1), "ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 1),
           "mfma_frac": round(flops / ms / 1e9 / PEAK_BF16_TFLOPS, 4)}
    del vis, aud
    torch.cuda.empty_cache()
    return out


def encoder_cpu_baseline():
    from oracle import imagebind_oracle as ib
    n = 8
    threads = torch.get_num_threads()
    st = ib.synthetic_state(ib.VISION_HUGE, seed=1234, init="fast")
    x = torch.randn(n, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    ib.vision_forward(x[:2], st)
    t0 = time.perf_counter()
    ib.vision_forward(x, st)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "frame-embeddings/s", "cores": threads, "kind": "port",
            "sample": f"fp32 torch-CPU oracle of the same ViT-H/14 tower (32 blocks), {n} synthetic frames, one timed "
                      f"pass after a 2-frame warm-up, torch intra-op threads = {threads}"}


def torch_rocm_reference():
    """The same ViT-H/14 tower through stock PyTorch-ROCm operators on this GPU (tools/torch_vit_probe.py): fp32 is what the
    reference runs (ImageBind.forward under no_grad, no autocast, foundation_models.py:116-133), bf16 is the vendor-library
    route.  A reported reference point, never part of the product path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("torch_vit_probe_lib", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                  "tools", "torch_vit_lib.py"))
    lib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lib)
    out = {}
    for name, dtype, iters in (("fp32", torch.float32, 2), ("bf16", torch.bfloat16, 4)):
        ms = lib.time_forward(FRAMES_PER_GPU, dtype, iters)
        out[name] = {"ms_per_forward": round(ms, 2), "frames_per_s": round(FRAMES_PER_GPU / ms * 1e3, 1)}
    out["what"] = ("stock PyTorch-ROCm operators (vendor GEMM, scaled_dot_product_attention, layer_norm, gelu), same batch, "
                   "random weights; fp32 = the reference's own execution mode")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scan", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    n_gpus = world

    from hippomm_amd.consolidation import select_key_frames_device
    from hippomm_amd.encoder import HipTower, synthetic_state_dict
    from hippomm_amd.sharding import all_gather_embeddings

    sd = synthetic_state_dict(("vision",), seed=1234)
    tower = HipTower("vision", sd)
    del sd
    torch.cuda.empty_cache()
    frames = torch.randn(FRAMES_PER_GPU, 3, 224, 224, device="cuda",
                         generator=torch.Generator(device="cuda").manual_seed(rank))
    emb = torch.empty(FRAMES_PER_GPU, 1024, dtype=torch.float32, device="cuda")
    counts = [FRAMES_PER_GPU] * world

    def step():
        tower.forward_into(frames, emb)
        feats = all_gather_embeddings(emb, counts)
        return select_key_frames_device(feats)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kept = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = n_gpus * FRAMES_PER_GPU * args.steps
        value = total_frames / elapsed
        ms_step = elapsed / args.steps * 1e3
        enc_flops = tower.flops(FRAMES_PER_GPU)
        roof, table = gemm_roofline(FRAMES_PER_GPU * 257)
        line = {
            "metric": "frame-embeddings/sec (ImageBind-huge ViT-H/14 vision tower; scan GB/s under 'scan')",
            "value": round(value, 1), "unit": "frame-embeddings/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE cfg2: ViT-H/14 vision encoder (32 blocks, 257 tokens, D=1280), 256 "
                                   "synthetic 224x224 frames per GPU per step, random-init weights; step = encode -> "
                                   "(N>1: RCCL all-gather of embeddings) -> cosine key-frame selection",
                       "frames_per_gpu": FRAMES_PER_GPU, "sharding": f"frames x{n_gpus}",
                       "kept_key_frames": int(kept.numel())},
            "step_tflops": round(enc_flops * args.steps / elapsed / 1e12, 1),
            "step_mfma_frac": round(enc_flops * args.steps / elapsed / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": roof, "kernels": table,
        }
        if n_gpus == 1:
            del tower
            torch.cuda.empty_cache()
            if not args.no_scan:
                line["scan"] = scan_bench(do_cpu=not args.no_cpu_baseline)
                line["joint_vision_audio"] = joint_bench()
            if not args.no_cpu_baseline:
                line["torch_rocm_reference"] = torch_rocm_reference()
                line["cpu_baseline"] = encoder_cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
