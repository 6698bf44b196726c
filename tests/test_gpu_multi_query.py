"""Batched feature_search: hmm_cosine_topk_multi against the oracle, query by query, and against the
single-query scan."""
import numpy as np
import pytest
import torch

from oracle.vector_ops_oracle import top_k_documented_order
from tests.test_gpu_scan import assert_topk_matches

pytestmark = pytest.mark.gpu


def _unit_rows(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, 1024)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("n,nq,k", [(5000, 3, 5), (4097, 16, 32), (70000, 7, 5), (300, 16, 5), (1, 2, 5), (255, 1, 64),
                                    (20000, 33, 8), (9000, 17, 64), (257, 5, 1), (513, 16, 3)])
def test_multi_query_matches_oracle(n, nq, k):
    from hippomm_amd.vector_ops import FeatureStore
    store = _unit_rows(n, n + nq)
    rng = np.random.default_rng(k)
    queries = rng.standard_normal((nq, 1024)).astype(np.float32)
    queries[0] = store[n // 2] + 0.2 * queries[0] / 32                    # one query close to a stored row
    fs = FeatureStore(store)
    res = fs.search_multi(queries, k)
    assert len(res) == nq
    for qi, (idx, sims) in enumerate(res):
        assert idx.dtype == np.int64 and len(idx) == min(k, n)
        assert_topk_matches(idx, sims, store, queries[qi], k)
    assert res[0][0][0] == n // 2


def test_multi_query_equals_single_query_scan_on_separated_data():
    """Clustered store: the top similarities are far apart, so the batched and the single-query kernels must agree
    exactly on the indices (and to fp32 rounding on the values)."""
    from hippomm_amd.vector_ops import FeatureStore
    rng = np.random.default_rng(3)
    n = 30000
    store = _unit_rows(n, 9)
    queries = np.stack([store[i] + 0.05 * rng.standard_normal(1024).astype(np.float32) / 32 for i in (7, 511, 29999, 12345)])
    fs = FeatureStore(store)
    multi = fs.search_multi(queries, 5)
    for qi in range(4):
        idx, sims = fs.search(queries[qi], 5)
        assert multi[qi][0][0] == (7, 511, 29999, 12345)[qi]
        assert multi[qi][0][0] == idx[0]
        np.testing.assert_allclose(multi[qi][1], sims, atol=2e-6)


def test_multi_query_ties_nan_rows_and_large_k_fallback():
    from hippomm_amd.vector_ops import FeatureStore
    store = _unit_rows(600, 4)
    store[100] = store[50]                                  # exact tie: higher row first
    store[200] = 0.0                                        # zero row -> NaN similarity -> rank 1 (reference quirk)
    q = np.stack([store[50] * 3.0, _unit_rows(1, 8)[0]])
    fs = FeatureStore(store)
    (i0, s0), (i1, s1) = fs.search_multi(q, 4)
    assert i0[0] == 200 and np.isnan(s0[0]) and i0[1:3].tolist() == [100, 50]
    assert i1[0] == 200 and np.isnan(s1[0])
    with np.errstate(invalid="ignore", divide="ignore"):
        all_sims = (store @ q[0]) / (np.linalg.norm(store, axis=1) * np.linalg.norm(q[0]))
    all_sims[100] = all_sims[50]                            # the tie is exact by construction
    w0, _ = top_k_documented_order(all_sims, 4)             # numpy's own tie order is unstable; ours is documented
    assert i0.tolist() == w0.tolist()
    # k > 64: per-query fallback inside the library
    clean = _unit_rows(700, 5)
    q2 = _unit_rows(3, 6)
    big = FeatureStore(clean).search_multi(q2, 100)
    for qi in range(3):
        assert_topk_matches(big[qi][0], big[qi][1], clean, q2[qi], 100)


def test_multi_query_argument_errors():
    from hippomm_amd import _lib
    from hippomm_amd.vector_ops import FeatureStore
    fs = FeatureStore(_unit_rows(100, 1))
    with pytest.raises(ValueError):
        fs.search_multi_device(torch.zeros(2, 512, device="cuda"), 5)
    lib = _lib.load()
    q = torch.zeros(2, 1024, device="cuda")
    out_i = torch.empty(2, 5, dtype=torch.int64, device="cuda"); out_s = torch.empty(2, 5, device="cuda")
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")
    rc = lib.hmm_cosine_topk_multi(fs.rows.data_ptr(), 100, 1024, q.data_ptr(), 2, 5, out_i.data_ptr(), out_s.data_ptr(),
                                   None, ws.data_ptr(), 16, _lib.stream_ptr())
    assert rc != 0 and b"workspace" in lib.hmm_last_error()
