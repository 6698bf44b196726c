#!/usr/bin/env python3
"""One burst of 32 frames (the reference's frame buffer) on T decode threads: wall time until the last frame is in the ring and, per
frame, how long the read / the libjpeg call / the pack took -- which phase inflates when more threads run than the quota has CPUs?
And the same burst on forked worker processes (no torch, no GPU in this script: forking is safe here).
    python tools/decode_burst_probe.py [out.json]"""
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
from PIL import Image

H, W = 720, 1280


def make_files(folder, n=32):
    rng = np.random.default_rng(0)
    base = np.asarray(Image.fromarray(rng.integers(0, 255, (6, 8, 3), dtype=np.uint8)).resize((W, H), Image.BICUBIC)).astype(np.float32)
    paths = []
    for i in range(n):
        p = os.path.join(folder, f"f{i:03d}.jpg")
        Image.fromarray(np.clip(base + rng.normal(0, 4, (H, W, 3)), 0, 255).astype(np.uint8)).save(p, quality=90)
        paths.append(p)
    return paths


def proc_decode(path):
    t0 = time.perf_counter()
    with open(path, "rb") as fh:
        data = fh.read()
    im = Image.core.new("RGB", (W, H))
    dec = Image._getdecoder("RGB", "jpeg", ("RGB", ""), ())
    dec.setimage(im, (0, 0, W, H))
    dec.decode(data)
    dec.cleanup()
    return time.perf_counter() - t0


def main():
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "hippomm_amd", "libhippomm_hip.so")) if False else None
    from concurrent.futures import ThreadPoolExecutor, ProcessPoolExecutor
    import multiprocessing as mp
    import importlib.util
    # the package's decode helpers without importing torch: load preprocess.py's functions through a stub `torch`
    folder = tempfile.mkdtemp(prefix="hmm_burst_")
    paths = make_files(folder)
    out = {"cpu.max": open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, "threads": {}, "procs": {}}
    from hippomm_amd import preprocess as pp, _lib
    hlib = _lib.load()
    win = pp.needed_window(H, W)
    ring = np.zeros((32, win[3], win[2], 3), np.uint8)

    def work(i):
        t0 = time.perf_counter()
        with open(paths[i], "rb") as fh:
            data = fh.read()
        t1 = time.perf_counter()
        ok = pp._decode_jpeg_direct(data, W, H, ring[i], hlib, win)
        t2 = time.perf_counter()
        return t0, t1, t2, ok

    for T in (1, 4, 8, 14, 16, 20, 24, 32):
        with ThreadPoolExecutor(T) as pool:
            list(pool.map(work, range(32)))
            walls, reads, decs, starts = [], [], [], []
            for _ in range(7):
                time.sleep(0.05)
                t_begin = time.perf_counter()
                res = list(pool.map(work, range(32)))
                walls.append((time.perf_counter() - t_begin) * 1e3)
                reads += [(r[1] - r[0]) * 1e3 for r in res]
                decs += [(r[2] - r[1]) * 1e3 for r in res]
                starts.append(max(r[0] for r in res) - t_begin)
            out["threads"][T] = {"wall_ms_median": round(statistics.median(walls), 2), "wall_ms_min": round(min(walls), 2),
                                 "read_ms_mean": round(statistics.mean(reads), 3), "decode_pack_ms_mean": round(statistics.mean(decs), 3),
                                 "decode_pack_ms_max": round(max(decs), 3), "last_start_ms": round(statistics.median(starts) * 1e3, 2)}
            print(T, json.dumps(out["threads"][T]), flush=True)
    for P in (8, 14, 16, 32):
        with ProcessPoolExecutor(P, mp_context=mp.get_context("fork")) as pool:
            list(pool.map(proc_decode, paths))
            walls, per = [], []
            for _ in range(7):
                time.sleep(0.05)
                t_begin = time.perf_counter()
                res = list(pool.map(proc_decode, paths))
                walls.append((time.perf_counter() - t_begin) * 1e3)
                per += [r * 1e3 for r in res]
            out["procs"][P] = {"wall_ms_median": round(statistics.median(walls), 2), "wall_ms_min": round(min(walls), 2),
                               "decode_ms_mean": round(statistics.mean(per), 3), "decode_ms_max": round(max(per), 3)}
            print("procs", P, json.dumps(out["procs"][P]), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
    import shutil
    shutil.rmtree(folder, ignore_errors=True)


if __name__ == "__main__":
    main()
