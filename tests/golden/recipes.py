"""Seeded input recipes shared by make_golden.py (which runs the reference on them
in the build container) and by the tests (which regenerate the same inputs and
compare against the committed expected outputs).  Each golden file stores the
SHA-256 of the input bytes so that a drift in the generator is detected rather
than silently compared against stale expectations.
"""
from __future__ import annotations

import hashlib

import numpy as np

D = 1024


def sha256(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# ------------------------------------------------------------------ scan (a9)
def scan_case(name: str):
    """-> (query, store, k) for a named top-k case."""
    if name in ("n4096_k5", "n4096_k32", "n4096_k32_f64store"):
        rng = np.random.default_rng(1234)
        store = rng.standard_normal((4096, D), dtype=np.float32)
        query = rng.standard_normal(D, dtype=np.float32)
        if name.endswith("f64store"):
            store = store.astype(np.float64)      # dtype after load_theta_event (:391-395)
        return query, store, (5 if name == "n4096_k5" else 32)
    if name == "n300_k5_unitrows":
        # shape of a real event: a few hundred unit-norm frame embeddings, k=5 (:3153)
        rng = np.random.default_rng(77)
        store = rng.standard_normal((300, D)).astype(np.float32)
        store /= np.linalg.norm(store, axis=1, keepdims=True)
        query = (store[17] + 0.3 * rng.standard_normal(D)).astype(np.float32)
        return query, store, 5
    if name in ("n1000_k5_unitrows_audio20", "n2500_k5_unitrows", "n64_k5_event", "n777_k32_unitrows_f64"):
        # more event-shaped cases (hippocampal_memory.py:3153, :3304): unit-norm vision rows / audio rows of length <= 20
        # (mean of three 20 x unit vectors), the query a noisy copy of a stored row, k = 5; one float64 store as after
        # load_theta_event (:391-395)
        n = {"n1000_k5_unitrows_audio20": 1000, "n2500_k5_unitrows": 2500, "n64_k5_event": 64, "n777_k32_unitrows_f64": 777}[name]
        rng = np.random.default_rng(1000 + n)
        store = rng.standard_normal((n, D)).astype(np.float32)
        store /= np.linalg.norm(store, axis=1, keepdims=True)
        if name == "n1000_k5_unitrows_audio20":
            others = rng.standard_normal((2, n, D)).astype(np.float32)
            others /= np.linalg.norm(others, axis=2, keepdims=True)
            store = ((store + 0.3 * others[0] + 0.3 * others[1]) * np.float32(20.0 / 3.0)).astype(np.float32)
        query = (store[n // 3] + 0.25 * rng.standard_normal(D).astype(np.float32) * np.linalg.norm(store[n // 3]) / 32).astype(np.float32)
        if name.endswith("f64"):
            store = store.astype(np.float64)
        return query, store, (32 if "k32" in name else 5)
    if name in ("n32_k5_textquery", "n150_k5_textquery", "n450_k5_textquery", "n3600_k5_textquery_f64"):
        # the retrieval call as QARecallSystem makes it (hippocampal_memory.py:2173-2176 -> :3153): the query is a TEXT embedding
        # (unit vector x exp(log_logit_scale) = 1 / 0.07), the store one event's unit-norm vision rows (32 = a frame buffer,
        # 3600 = an hour at 1 fps), k = 5; weakly related rows, i.e. the small similarity gaps of a real cross-modal query
        n = int(name.split("_")[0][1:])
        rng = np.random.default_rng(4000 + n)
        store = rng.standard_normal((n, D)).astype(np.float32)
        store /= np.linalg.norm(store, axis=1, keepdims=True)
        target = store[(2 * n) // 3] + store[n // 5]
        query = target + 2.0 * rng.standard_normal(D).astype(np.float32) * np.linalg.norm(target) / 32
        query = (query / np.linalg.norm(query) * np.float32(1.0 / 0.07)).astype(np.float32)
        if name.endswith("f64"):
            store = store.astype(np.float64)
        return query, store, 5
    if name in ("k_zero", "k_negative", "k_more_negative_than_n", "empty_store"):
        # what the reference's slice argsort(sims)[-k:][::-1] (vector_ops.py:185) does outside 1 <= k: k = 0 returns ALL rows
        # ([-0:] is the whole array), k < 0 the best N - |k| rows, nothing when |k| >= N; an empty store returns nothing
        rng = np.random.default_rng(14)
        store = rng.standard_normal((9, D), dtype=np.float32)
        query = rng.standard_normal(D, dtype=np.float32)
        if name == "empty_store":
            return query, np.zeros((0, D), np.float32), 5
        return query, store, {"k_zero": 0, "k_negative": -3, "k_more_negative_than_n": -20}[name]
    if name == "k_gt_n":
        rng = np.random.default_rng(5)
        return rng.standard_normal(D, dtype=np.float32), rng.standard_normal((7, D), dtype=np.float32), 32
    if name == "store_1d":
        rng = np.random.default_rng(6)
        return rng.standard_normal(D, dtype=np.float32), rng.standard_normal(D, dtype=np.float32), 5
    if name == "duplicate_rows":
        rng = np.random.default_rng(8)
        store = rng.standard_normal((12, D), dtype=np.float32)
        query = (store[3] + 0.1 * rng.standard_normal(D, dtype=np.float32)).astype(np.float32)
        store[9] = store[3]                         # exact tie at rank 1/2
        store[5] = store[3]
        return query, store, 4
    if name == "zero_row":
        rng = np.random.default_rng(9)
        store = rng.standard_normal((10, D), dtype=np.float32)
        store[4] = 0.0                              # 0/0 -> NaN similarity
        return rng.standard_normal(D, dtype=np.float32), store, 3
    raise KeyError(name)


SCAN_CASES = ["n4096_k5", "n4096_k32", "n4096_k32_f64store", "n300_k5_unitrows",
              "k_gt_n", "store_1d", "duplicate_rows", "zero_row",
              "n1000_k5_unitrows_audio20", "n2500_k5_unitrows", "n64_k5_event", "n777_k32_unitrows_f64",
              "n32_k5_textquery", "n150_k5_textquery", "n450_k5_textquery", "n3600_k5_textquery_f64",
              "k_zero", "k_negative", "k_more_negative_than_n", "empty_store"]


# ------------------------------------------------------------- select (a7)
def clustered(n: int, n_clusters: int, sigma: float, seed: int) -> np.ndarray:
    """Time-ordered frames that dwell on `n_clusters` scenes: consecutive runs of
    near-duplicates (centre + sigma * noise), so that consolidation drops frames."""
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((n_clusters, D)).astype(np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    which = (np.arange(n) * n_clusters) // n
    noise = rng.standard_normal((n, D)).astype(np.float32) * np.float32(sigma / np.sqrt(D))
    return (centres[which] + noise).astype(np.float32)


def select_case(name: str):
    """-> (features fp32 (n,1024), times float64 (n,))."""
    if name in ("n1", "n2", "n3"):
        n = int(name[1:])
        f = np.random.default_rng(20 + n).standard_normal((n, D), dtype=np.float32)
    elif name == "n32_clusters6":
        f = clustered(32, 6, 0.2, 7)                 # BASELINE cfg 1 shape
    elif name == "n257_clusters40":
        f = clustered(257, 40, 0.25, 11)
    elif name == "n3600_clusters600":
        f = clustered(3600, 600, 0.2, 13)            # BASELINE cfg 5 shape
    elif name == "n64_revisit":
        # scenes revisited later in time: the kept set must block late near-duplicates
        base = clustered(16, 4, 0.15, 31)
        f = np.concatenate([base, base[::-1] + np.float32(1e-3), base, base[::2].repeat(2, 0)])
    elif name == "n40_all_distinct":
        f = np.random.default_rng(41).standard_normal((40, D), dtype=np.float32)
    elif name == "n40_all_same":
        v = np.random.default_rng(42).standard_normal(D).astype(np.float32)
        f = np.tile(v, (40, 1)) * np.linspace(0.5, 2.0, 40, dtype=np.float32)[:, None]
    elif name == "n24_duplicates":
        f = clustered(24, 24, 0.0, 43)
        f[7] = f[2]
        f[15] = f[2]
        f[23] = f[11]
    elif name == "n20_near_threshold":
        # one pair placed 1e-4 either side of 0.9 (outside the BLAS-order band)
        rng = np.random.default_rng(44)
        f = rng.standard_normal((20, D)).astype(np.float32)
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        def at_cos(u, c, seed):
            r = np.random.default_rng(seed).standard_normal(D)
            r -= r.dot(u) * u
            r /= np.linalg.norm(r)
            return (c * u + np.sqrt(1 - c * c) * r).astype(np.float32)
        f[5] = at_cos(f[1].astype(np.float64), 0.9 + 1e-4, 1)
        f[9] = at_cos(f[3].astype(np.float64), 0.9 - 1e-4, 2)
    elif name in ("n24_inband_1e6", "n30_inband_band"):
        # pairs placed INSIDE / at the edge of the band where the reference's answer depends on its BLAS summation order
        # ("a row pair at cosine 0.9 +- 1e-6").  Offsets from float32(0.9): +-1e-6 (n24) and a ladder
        # +-3e-7, +-1e-7, +-3e-8, 0 (n30).  Row 2k+1 is the partner of row 2k; the other rows are unrelated.
        offs = [1e-6, -1e-6, 2e-6, -2e-6] if name == "n24_inband_1e6" else [3e-7, -3e-7, 1e-7, -1e-7, 3e-8, -3e-8, 0.0]
        n = 24 if name == "n24_inband_1e6" else 30
        rng = np.random.default_rng(46 if name == "n24_inband_1e6" else 47)
        f = rng.standard_normal((n, D)).astype(np.float32)
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        thr = float(np.float32(0.9))
        for j, off in enumerate(offs):
            u = f[2 * j].astype(np.float64)
            u /= np.linalg.norm(u)
            r = np.random.default_rng(100 + j).standard_normal(D)
            r -= r.dot(u) * u
            r /= np.linalg.norm(r)
            c = thr + off
            f[2 * j + 1] = ((c * u + np.sqrt(1 - c * c) * r) * (1.0 + 0.37 * j)).astype(np.float32)   # un-normalised partner
    elif name == "n12_zero_row":
        f = clustered(12, 5, 0.2, 45)
        f[6] = 0.0                                  # NaN row: never kept, blocks nobody later
    else:
        raise KeyError(name)
    n = f.shape[0]
    return np.ascontiguousarray(f, dtype=np.float32), np.arange(n, dtype=np.float64)


SELECT_CASES = ["n1", "n2", "n3", "n32_clusters6", "n257_clusters40", "n3600_clusters600",
                "n64_revisit", "n40_all_distinct", "n40_all_same", "n24_duplicates",
                "n20_near_threshold", "n12_zero_row"]


# In-band cases: NOT part of SELECT_CASES.  Inside ~1e-7 of the threshold the reference's own answer depends on the sgemm
# summation order of the host it runs on; the golden stored for these is informational, and what the tests pin is that the
# HIP path equals the fp64-accumulated definition (oracle select_key_frames_exact).
SELECT_INBAND_CASES = ["n24_inband_1e6", "n30_inband_band"]


# ------------------------------------------------------------- memory_store event (8f-2)
def event_case():
    """A small consolidated event: 3 vision rows (all frames), 2 audio rows, kept frames 0 and 2."""
    rng = np.random.default_rng(2024)
    vision = rng.standard_normal((3, D)).astype(np.float32)
    vision /= np.linalg.norm(vision, axis=1, keepdims=True)
    audio = (20.0 * rng.standard_normal((2, D)) / np.sqrt(D)).astype(np.float32)
    return dict(
        features={"vision": vision, "vision_times": np.array([0.0, 1.0, 2.5]),
                  "audio": audio, "audio_times": np.array([0.0, 10.0])},
        feature_times=None,
        frames=["frames/vid/t_0000/frame_000000.jpg", "frames/vid/t_0002/frame_000075.jpg"],
        frame_times=[0.0, 2.5],
        frame_captions=["a person opens a door", "the same person sits down"],
        audio_times=[0.0, 10.0],
        audio_transcription=[{"text": "hello there", "start": 0.2, "end": 1.1}],
        holistic_audio_transcription=[{"text": "hello there", "start": 0.2, "end": 1.1}],
        summary="Someone enters a room and sits down.",
        start_time=0.0, end_time=12.5)
