"""Oracle (test infrastructure, not product): greedy key-frame selection on the CPU.

Restates reference ``HippocampalMemory._select_key_frames``
(hippomm/core/hippocampal_memory.py:944-967) in numpy:

  :947-948  n <= 2            -> every index is kept
  :951      rows divided by their L2 norm (fp32 in, fp32 out)
  :952      S = Fn @ Fn.T     (sgemm)
  :955      kept = [0]
  :958-961  for i = 1..n-1: keep i iff every S[i, kept] < threshold
  :964-965  "always include last if distinct" -- kept for fidelity; it can
            never fire (if n-1 is already kept S[-1,-1] ~ 1 blocks it, otherwise
            some S[-1,k] >= threshold blocks it)
  :967      int64 array of kept indices (increasing, starts with 0)

The comparison happens in float32: ``S`` is float32 and numpy casts the Python
float 0.9 to float32(0.9) = 0.89999998 (numpy 2 weak scalars; numpy 1.26
value-based casting gives the same).  A NaN similarity (zero-norm row) is not
``< threshold`` and therefore blocks the candidate.

Pinned against the unmodified reference: tests/golden/select_*.json.

``select_key_frames_exact`` is the definition the HIP path implements: the same
fp32 normalisation, but every dot product accumulated in float64 and rounded to
float32 once before the comparison.  It differs from the sgemm version only for
entries within ~1e-7 of the threshold, where the reference itself depends on
the BLAS summation order of the machine it runs on.
"""
from __future__ import annotations

import numpy as np


def select_key_frames_oracle(features: np.ndarray, times=None,
                             similarity_threshold: float = 0.9) -> np.ndarray:
    n = len(features)
    if n <= 2:
        return np.arange(n)
    unit = features / np.linalg.norm(features, axis=1, keepdims=True)
    gram = np.dot(unit, unit.T)
    kept = [0]
    for i in range(1, n):
        if np.all(gram[i, kept] < similarity_threshold):
            kept.append(i)
    if n > 1 and np.all(gram[-1, kept] < similarity_threshold):
        kept.append(n - 1)
    return np.array(kept)


def evaluated_margin(features: np.ndarray, similarity_threshold: float = 0.9) -> float:
    """min |S[i,k] - thr| over the comparisons the greedy loop actually evaluates."""
    n = len(features)
    if n <= 2:
        return float("inf")
    unit = features / np.linalg.norm(features, axis=1, keepdims=True)
    gram = np.dot(unit, unit.T).astype(np.float64)
    thr = float(np.float32(similarity_threshold))
    kept = [0]
    margin = float("inf")
    for i in range(1, n):
        row = gram[i, kept]
        with np.errstate(invalid="ignore"):
            d = np.abs(row - thr)
        if d.size and np.isfinite(d).any():
            margin = min(margin, float(np.nanmin(d)))
        if np.all(row < thr):
            kept.append(i)
    return margin


def select_key_frames_exact(features: np.ndarray,
                            similarity_threshold: float = 0.9) -> np.ndarray:
    """fp32 normalise, float64 dot, round to fp32, compare -- the HIP definition."""
    n = len(features)
    if n <= 2:
        return np.arange(n)
    f32 = np.ascontiguousarray(features, dtype=np.float32)
    norm = np.sqrt(np.sum(f32.astype(np.float64) ** 2, axis=1)).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        unit = (f32 / norm[:, None]).astype(np.float32)
    u64 = unit.astype(np.float64)
    with np.errstate(invalid="ignore"):
        gram = (u64 @ u64.T).astype(np.float32)
    # dgemm need not be bitwise symmetric; the HIP gram is (it evaluates each
    # unordered pair once), so mirror the upper triangle.
    gram = np.triu(gram) + np.triu(gram, 1).T
    thr = np.float32(similarity_threshold)
    blocked = np.zeros(n, dtype=bool)
    kept = []
    for i in range(n):
        if i > 0 and blocked[i]:
            continue
        kept.append(i)
        with np.errstate(invalid="ignore"):
            blocked |= ~(gram[i] < thr)
    return np.array(kept, dtype=np.int64)


def pair_similarities(features: np.ndarray, n_pairs: int) -> np.ndarray:
    """float64 cosine of the fp32-normalised rows (2j, 2j+1), j < n_pairs, under the HIP definition (before the final
    rounding to fp32): how far the in-band fixtures' pairs really sit from the threshold."""
    f32 = np.ascontiguousarray(features, dtype=np.float32)
    norm = np.sqrt(np.sum(f32.astype(np.float64) ** 2, axis=1)).astype(np.float32)
    unit = (f32 / norm[:, None]).astype(np.float32).astype(np.float64)
    return np.array([unit[2 * j].dot(unit[2 * j + 1]) for j in range(n_pairs)])
