#!/usr/bin/env python3
"""Where does the host's decode throughput stop scaling?  (GPU box host; no GPU work.)

  cgroup  : cpu.max / cpu.stat of this container (a CFS quota caps the usable cores whatever nproc says)
  spin    : a GIL-free compute loop (sha256 of a cached 4 MB buffer) on T threads -> the cores really available
  decode  : Pillow decode + pack into a preallocated ring on T threads, with and without Pillow's block cache
            (Image.core.set_blocks_max) -> interpreter-lock and allocator effects
  procs   : the same decode in P forked worker processes (no shared interpreter lock, no shared address space)

    python tools/decode_scaling_probe.py [out.json]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
from PIL import Image

H, W = 720, 1280


def make_files(folder, n=64):
    rng = np.random.default_rng(0)
    paths = []
    base = np.asarray(Image.fromarray(rng.integers(0, 255, (6, 8, 3), dtype=np.uint8)).resize((W, H), Image.BICUBIC)).astype(np.float32)
    for i in range(n):
        frame = np.clip(base + rng.normal(0, 4, (H, W, 3)), 0, 255).astype(np.uint8)
        p = os.path.join(folder, f"f{i:03d}.jpg")
        Image.fromarray(frame).save(p, quality=90)
        paths.append(p)
    return paths


def run_threads(fn, threads, n):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(fn, range(2 * threads)))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            list(pool.map(fn, range(n)))
            best = min(best, time.perf_counter() - t0)
    return n / best


def child(mode, threads, n, folder):
    from hippomm_amd import _lib, preprocess as pp
    lib = _lib.load()
    paths = sorted(os.path.join(folder, f) for f in os.listdir(folder))
    if "blocks" in mode:
        Image.core.set_blocks_max(2 * threads + 8)
    ring = np.zeros((max(2 * threads, 64), H, W, 3), np.uint8)
    if mode.startswith("spin"):
        import hashlib
        src = bytes(4 << 20)

        def fn(i):
            hashlib.sha256(src).digest()                         # pure compute, interpreter lock released
        rate = run_threads(fn, threads, n)
    elif mode.startswith("decode"):
        def fn(i):
            with open(paths[i % len(paths)], "rb") as fh:
                im = pp._open_rgb(fh)
            if "nopack" not in mode:
                pp._pack_into(im, ring[i % len(ring)], lib)
        rate = run_threads(fn, threads, n)
    print(json.dumps({"mode": mode, "threads": threads, "per_s": round(rate, 1)}), flush=True)


def proc_worker(args):
    path, = args
    with open(path, "rb") as fh:
        im = Image.open(fh)
        im.load()
    return im.size[0]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        return
    out = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
        try:
            out[f] = open(f).read().strip()
        except OSError:
            pass
    print(json.dumps(out), flush=True)
    folder = tempfile.mkdtemp(prefix="hmm_dsp_")
    make_files(folder)
    rows = []
    for mode in ("spin", "decode", "decode_blocks", "decode_nopack_blocks"):
        for t in (1, 2, 4, 8, 16, 32, 64, 128):
            env = dict(os.environ)
            r = subprocess.run([sys.executable, __file__, "--child", mode, str(t), str(max(64, 4 * t)), folder], capture_output=True, text=True, env=env)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"mode": mode, "threads": t, "error": r.stderr[-300:]})
            print(line, flush=True)
            rows.append(json.loads(line))
    for var, val in (("MALLOC_ARENA_MAX", "1"), ("MALLOC_MMAP_THRESHOLD_", "268435456"), ("MALLOC_TOP_PAD_", "268435456")):
        for t in (16, 64):
            env = dict(os.environ, **{var: val})
            r = subprocess.run([sys.executable, __file__, "--child", "decode", str(t), str(4 * t), folder], capture_output=True, text=True, env=env)
            try:
                rec = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                rec = {"error": r.stderr[-300:]}
            rec["env"] = f"{var}={val}"
            print(json.dumps(rec), flush=True)
            rows.append(rec)
    # forked worker processes
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    paths = sorted(os.path.join(folder, f) for f in os.listdir(folder))
    for p in (8, 16, 32, 64, 128):
        with ProcessPoolExecutor(p, mp_context=mp.get_context("fork")) as pool:
            work = [(paths[i % len(paths)],) for i in range(4 * p)]
            list(pool.map(proc_worker, work[: 2 * p]))
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                list(pool.map(proc_worker, work, chunksize=2))
                best = min(best, time.perf_counter() - t0)
        rec = {"mode": "procs_decode_only", "procs": p, "per_s": round(len(work) / best, 1)}
        print(json.dumps(rec), flush=True)
        rows.append(rec)
    try:
        out["cpu.stat_after"] = open("/sys/fs/cgroup/cpu.stat").read().strip()
    except OSError:
        pass
    out["rows"] = rows
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
    import shutil
    shutil.rmtree(folder, ignore_errors=True)


if __name__ == "__main__":
    main()
