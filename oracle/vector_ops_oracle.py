"""Oracle (test infrastructure, not product): cosine top-k scan on the CPU.

Restates reference ``top_k_cosine_similarity``
(hippomm/utils/vector_ops.py:151-188) in numpy, operation for operation, so
that on the same machine/BLAS it returns what the reference returns:

  :166-169  torch inputs are brought to numpy
  :172-174  query flattened to 1-D, a 1-D store becomes one row
  :178-179  ||a||  and row norms ||b_i||   (np.linalg.norm)
  :182      sims = (b @ a) / (||b_i|| * ||a||)
  :185      order = argsort(sims) ascending, last k, reversed
  :186      sims gathered at those positions

Pinned against the unmodified reference: tests/golden/scan_*.json
(tests/test_oracle_golden.py).

``scan_order_key`` is the *documented* total order the HIP path implements for
the cases numpy leaves to its sort implementation (exact ties, NaN):
NaN ranks above every number (numpy sorts NaN last, the reference then
reverses), and among equal keys the HIGHER row index comes first (what a
stable ascending sort followed by the reference's reversal yields).  -0.0 and
+0.0 compare equal.
"""
from __future__ import annotations

import numpy as np

try:  # torch is optional for the oracle; the reference accepts tensors
    import torch
except Exception:  # pragma: no cover
    torch = None


def _as_numpy(x):
    if torch is not None and isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return x


def top_k_cosine_similarity_oracle(a, b, k):
    """(indices int64[k'], sims[k']) with k' = min(k, N); see module docstring."""
    query = _as_numpy(a)
    store = _as_numpy(b)
    query = query.reshape(-1)
    if store.ndim == 1:
        store = store.reshape(1, -1)
    query_len = np.linalg.norm(query)
    row_len = np.linalg.norm(store, axis=1)
    sims = np.dot(store, query) / (row_len * query_len)
    order = np.argsort(sims)[-k:][::-1]
    return order, sims[order]


def scan_order_key(sims: np.ndarray) -> np.ndarray:
    """uint64 key whose descending order is the HIP path's documented order.

    key = (ordered-float32 bits << 32) | row index.  The float map is the usual
    monotone one (flip all bits of negatives, set the sign bit of positives),
    with every NaN sent to 0xFFFFFFFF and -0.0 canonicalised to +0.0.
    """
    s = np.asarray(sims, dtype=np.float32) + np.float32(0.0)
    bits = s.view(np.uint32).astype(np.uint64)
    neg = (bits >> np.uint64(31)) != 0
    mapped = np.where(neg, (~bits) & np.uint64(0xFFFFFFFF), bits | np.uint64(0x80000000))
    mapped = np.where(np.isnan(s), np.uint64(0xFFFFFFFF), mapped)
    idx = np.arange(s.shape[0], dtype=np.uint64)
    return (mapped << np.uint64(32)) | idx


def top_k_documented_order(sims: np.ndarray, k: int):
    """Top-k under ``scan_order_key`` (ties -> higher index first, NaN first)."""
    key = scan_order_key(sims)
    n = key.shape[0]
    k_eff = n if k == 0 else (max(n + k, 0) if k < 0 else min(k, n))      # the reference's slice [-k:] (vector_ops.py:185)
    order = np.argsort(key, kind="stable")[::-1][:k_eff]
    return order.astype(np.int64), np.asarray(sims)[order]
