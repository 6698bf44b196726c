"""Seeded random cases shared by tests/golden/live_reference.py (runs the UNMODIFIED reference, build container only) and
tests/test_oracle_vs_reference_live.py (runs the oracle): both sides rebuild the inputs from the seed, only results travel."""
import numpy as np

N_SCAN, N_SELECT = 60, 60


def scan_case(seed: int):
    """-> (query (1024,), store (n,1024) or (1024,), k).  Mixes dtypes, k > n, 1-D stores, duplicate rows (ties) and zero rows
    (NaN similarity: the reference ranks them first, vector_ops.py:182-185)."""
    r = np.random.default_rng(10_000 + seed)
    n = int(r.integers(1, 2500))
    k = int(r.integers(1, 48))
    dtype = np.float64 if seed % 3 == 0 else np.float32
    store = r.standard_normal((n, 1024)).astype(dtype)
    if seed % 4 == 1:
        store /= np.linalg.norm(store, axis=1, keepdims=True)          # unit rows, as events hold them
    if seed % 5 == 2 and n > 4:
        src = r.integers(0, n, size=max(1, n // 10))
        dst = r.integers(0, n, size=src.size)
        store[dst] = store[src]                                         # exact duplicates -> exact ties
    if seed % 7 == 3 and n > 2:
        store[r.integers(0, n, size=2)] = 0.0                           # zero rows -> NaN
    if seed % 11 == 4:
        store = store[0]                                                # 1-D b (vector_ops.py:172-174)
    q = r.standard_normal(1024).astype(np.float32)
    return q, store, k


def select_case(seed: int):
    """-> (features (n,1024) float32, times (n,), threshold).  Clustered rows so that the greedy rule keeps and drops frames;
    some seeds put duplicates and near-threshold pairs in."""
    r = np.random.default_rng(20_000 + seed)
    n = int(r.integers(1, 500))
    n_clusters = int(r.integers(1, max(2, n // 3 + 1)))
    centres = r.standard_normal((n_clusters, 1024))
    labels = np.sort(r.integers(0, n_clusters, size=n)) if seed % 2 == 0 else r.integers(0, n_clusters, size=n)
    noise = float(r.choice([0.05, 0.2, 0.33, 0.5]))
    f = (centres[labels] + noise * r.standard_normal((n, 1024))).astype(np.float32)
    if seed % 5 == 1 and n > 3:
        f[r.integers(0, n)] = f[r.integers(0, n)]                       # an exact duplicate
    thr = 0.9 if seed % 4 else float(r.choice([0.5, 0.8, 0.95]))
    return f, np.arange(n, dtype=np.float64), thr
