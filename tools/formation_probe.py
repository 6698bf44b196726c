#!/usr/bin/env python3
"""Probe of the files -> embeddings pipeline (hippomm_amd.preprocess.vision_pipeline) on the GPU box's host:

  a. decode alone: Pillow + the GIL-free packing call on 1 / 8 / 16 / 32 / 64 / 128 threads (frames/s), and the old way
     (np.asarray(Image.open().convert('RGB')) per frame on 8 threads) for reference;
  b. the whole call (paths -> embeddings on the host) for 32 and 256 frames per (workers, first_chunk, upload_min, depth);
  c. the audio call for 1 / 16 wav files (if --audio).

    python tools/formation_probe.py [out.json] [--audio]
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from hippomm_amd import preprocess as pp
from hippomm_amd.encoder import HipTower, synthetic_state_dict


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def old_decode(paths, workers=8):
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image

    def one(path):
        with open(path, "rb") as fh:
            return np.asarray(Image.open(fh).convert("RGB"), dtype=np.uint8)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return list(pool.map(one, paths))


def main():
    out_path = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    res = {"host_cpus": len(os.sched_getaffinity(0)), "torch_threads": torch.get_num_threads()}
    folder = tempfile.mkdtemp(prefix="hmm_probe_frames_")
    paths, mean_bytes = bench.write_synthetic_jpegs(folder, 256)
    res["jpeg_mean_bytes"] = int(mean_bytes)
    # ---- a. decode alone
    dec = {}
    for w in (1, 8, 14):
        if w > 2 * res["host_cpus"]:
            continue
        n = 16 if w == 1 else 256
        pp.decode_rgb(paths[:n], workers=w)
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            pp.decode_rgb(paths[:n], workers=w)
            t.append(time.perf_counter() - t0)
        dec[f"threads_{w}"] = {"frames": n, "ms": round(median(t) * 1e3, 2), "frames_per_s": round(n / median(t), 1)}
    t0 = time.perf_counter()
    old_decode(paths, 8)
    dec["round5_way_8_threads"] = {"frames": 256, "ms": round((time.perf_counter() - t0) * 1e3, 2)}
    dec["cpu_quota"] = pp.cpu_quota()
    dec["decode_workers_default"] = pp.decode_workers()
    res["decode_alone"] = dec
    print(json.dumps(dec), flush=True)
    # ---- b. the whole call
    tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
    rows = []
    for n in (32, 256):
        ps = paths[:n]
        emb = torch.empty(n, 1024, device="cuda")
        x = pp.load_and_transform_vision_data_device(ps, "cuda")
        fwd = bench.event_time_ms(lambda: tower.forward_into(x, emb), 5, warmup=2)
        want = emb.cpu()
        grid = [(0, 0, 8, 1, 0.0003, 0), (0, 0, 8, 1, 0.001, 0), (0, 0, 8, 1, 0.0003, 12), (0, 0, 8, 1, 0.001, 12), (0, 4, 8, 1, 0.001, 12),
                (0, 0, 8, 1, 0.0003, 0), (0, 0, 8, 1, 0.001, 0), (0, 0, 8, 1, 0.0003, 12), (0, 0, 8, 1, 0.001, 12), (0, 4, 8, 1, 0.001, 12)]
        for w, mc, um, mi, poll, tw in grid:
            if w > 2 * res["host_cpus"]:
                continue
            stats = {}

            def call():
                pp.vision_pipeline(ps, "cuda", lambda xx, lo, hi: tower.forward_into(xx[lo:hi], emb[lo:hi]), workers=w, first_chunk=mc,
                                   upload_min=um, depth=mi, stats=stats, poll_s=poll, tail_wait=tw)
                return emb.detach().cpu().numpy()
            call(); call()
            t = []
            for _ in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                got = call()
                t.append((time.perf_counter() - t0) * 1e3)
            rows.append({"frames": n, "workers": w, "first_chunk": mc, "upload_min": um, "depth": mi, "poll_s": poll, "tail_wait": tw, "ms": round(median(t), 3),
                         "ms_min": round(min(t), 3), "frames_per_s": round(n / median(t) * 1e3, 1), "ms_tensor_in_forward": round(fwd, 3),
                         "ranges": stats.get("chunks"), "same_bits": bool(np.array_equal(got, want.numpy()))})
            print(json.dumps(rows[-1]), flush=True)
    res["whole_call"] = rows
    # one traced call per setting: where the time goes
    traces = {}
    for w, mc in ((0, 0),):
        stats = {"trace": True}
        emb = torch.empty(256, 1024, device="cuda")
        for _ in range(3):
            stats = {"trace": True}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pp.vision_pipeline(paths, "cuda", lambda xx, lo, hi: tower.forward_into(xx[lo:hi], emb[lo:hi]), workers=w, first_chunk=mc, stats=stats)
            stats["issue_done_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            emb.cpu()
            stats["call_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        traces[f"workers_{w}_first_chunk_{mc}"] = stats
        print(json.dumps(stats), flush=True)
    res["traces"] = traces
    del tower
    torch.cuda.empty_cache()
    if "--bench-legs" in sys.argv:
        res["formation_from_files"] = bench.formation_bench(True)
        print(json.dumps(res["formation_from_files"]), flush=True)
    if "--audio" in sys.argv:
        res["audio"] = bench.audio_bench()
        print(json.dumps(res["audio"]), flush=True)
    if out_path:
        with open(out_path, "w") as fh:
            json.dump(res, fh, indent=1)
    import shutil
    shutil.rmtree(folder, ignore_errors=True)


if __name__ == "__main__":
    main()
