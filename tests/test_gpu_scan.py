"""GPU parity: hmm_cosine_topk (through hippomm_amd.vector_ops) vs the reference's golden
vectors, vs the numpy oracle on seeded inputs, and size-independent properties at the
BASELINE size (1M x 1024)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import recipes
from oracle.vector_ops_oracle import top_k_cosine_similarity_oracle, top_k_documented_order

pytestmark = pytest.mark.gpu

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "scan_golden.json").read_text())["cases"]
SIM_ATOL = 2e-6          # fp32 scan vs the reference's fp32/fp64 numpy result


def assert_topk_matches(idx, sims, store, q, k):
    """Indices must equal the oracle's wherever the oracle's neighbouring similarities are
    further apart than the fp32 tolerance (always the case for the reference's k=5 / cfg-4's
    k=32 on these inputs); inside a run of near-equal values (only seen when ranking thousands
    of rows) the fp32 scan may order them differently, so there the check is on values: the
    row returned at rank i must have the oracle's rank-i similarity within tolerance."""
    want_idx, want_sims = top_k_cosine_similarity_oracle(q, store, k)
    with np.errstate(invalid="ignore", divide="ignore"):
        all_sims = (store @ q) / (np.linalg.norm(store, axis=1) * np.linalg.norm(q))
    assert len(idx) == len(want_idx) and len(set(idx.tolist())) == len(idx)
    np.testing.assert_allclose(sims, want_sims, rtol=0, atol=SIM_ATOL)
    np.testing.assert_allclose(all_sims[idx], want_sims, rtol=0, atol=SIM_ATOL)
    # ranks whose oracle similarity is further than BAND from both neighbours must match exactly
    # (BAND = 1e-6 is ~5x the fp32 rounding noise of a 1024-term dot product on unit-scale rows)
    BAND = 1e-6
    kk = len(want_idx)
    ordered = np.sort(all_sims[~np.isnan(all_sims)])[::-1][: kk + 1].astype(np.float64)
    gaps = np.full(kk + 1, np.inf)
    gaps[1: len(ordered)] = ordered[:-1] - ordered[1:]              # gaps[i] = value[i-1] - value[i]
    separated = (gaps[:kk] > BAND) & (gaps[1: kk + 1] > BAND)
    assert np.array_equal(idx[separated], want_idx[separated])
    return int(separated.sum())


def _sims(case):
    return np.array([np.nan if s is None else s for s in case["sims"]], dtype=np.float64)


@pytest.mark.parametrize("name", recipes.SCAN_CASES)
def test_golden_vectors(name):
    from hippomm_amd.vector_ops import top_k_cosine_similarity
    case = GOLD[name]
    q, store, k = recipes.scan_case(name)
    assert recipes.sha256(q, store) == case["input_sha256"]
    idx, sims = top_k_cosine_similarity(q, store, k)
    assert idx.dtype == np.int64 and idx.tolist() == case["indices"]
    assert str(sims.dtype) == case["sims_dtype"]
    np.testing.assert_allclose(sims, _sims(case), rtol=0, atol=SIM_ATOL, equal_nan=True)


@pytest.mark.parametrize("n,k", [(1, 1), (1, 5), (2, 5), (63, 5), (4096, 32), (4097, 32), (8191, 7),
                                 (20000, 1), (20000, 5), (20000, 100), (20000, 1024), (9000, 2000),
                                 (5000, 5000), (70001, 32),
                                 # fused scan: one-kernel finish (k*k <= 4096) vs chunk passes (k in 65..128),
                                 # few scan blocks, a single straggler row, sims-buffer path (k > 128)
                                 (4097, 1), (4098, 64), (4104, 65), (12345, 128), (12345, 129), (300001, 64),
                                 (300001, 100)])
def test_matches_oracle(n, k):
    from hippomm_amd.vector_ops import top_k_cosine_similarity
    rng = np.random.default_rng(n * 31 + k)
    store = rng.standard_normal((n, 1024), dtype=np.float32)
    q = rng.standard_normal(1024, dtype=np.float32)
    idx, sims = top_k_cosine_similarity(q, store, k)
    n_exact = assert_topk_matches(idx, sims, store, q, k)
    assert n_exact >= 0.8 * len(idx)          # the tolerance band is the exception, not the rule


def test_accepts_torch_and_resident_store():
    from hippomm_amd.vector_ops import FeatureStore, top_k_cosine_similarity
    rng = np.random.default_rng(3)
    store = rng.standard_normal((1500, 1024), dtype=np.float32)
    q = rng.standard_normal(1024, dtype=np.float32)
    want_idx, want_sims = top_k_cosine_similarity_oracle(q, store, 5)
    for a, b in [(torch.from_numpy(q).reshape(1, 1024), torch.from_numpy(store)),
                 (torch.from_numpy(q).cuda(), torch.from_numpy(store).cuda()),
                 (q, FeatureStore(store))]:
        idx, sims = top_k_cosine_similarity(a, b, 5)
        assert idx.tolist() == want_idx.tolist()
        np.testing.assert_allclose(sims, want_sims, rtol=0, atol=SIM_ATOL)


def test_many_ties_and_nans_follow_documented_order():
    from hippomm_amd.vector_ops import top_k_cosine_similarity
    rng = np.random.default_rng(11)
    base = rng.standard_normal((50, 1024), dtype=np.float32)
    store = base[rng.integers(0, 50, size=6000)]          # every row duplicated ~120 times
    store[[17, 4000, 5999]] = 0.0                          # NaN rows
    q = rng.standard_normal(1024, dtype=np.float32)
    idx, sims = top_k_cosine_similarity(q, store, 300)
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = (store @ q) / (np.linalg.norm(store, axis=1) * np.linalg.norm(q))
    assert idx[:3].tolist() == [5999, 4000, 17] and np.isnan(sims[:3]).all()
    # GPU sims differ from numpy's in the last bits, so check the rule on the GPU's own values:
    # equal sims -> descending index; overall non-increasing.
    s = sims[3:]
    assert np.all(s[:-1] >= s[1:])
    eq = s[:-1] == s[1:]
    assert np.all(idx[3:][:-1][eq] > idx[3:][1:][eq])
    np.testing.assert_allclose(s, ref[idx[3:]], rtol=0, atol=SIM_ATOL)


def test_zero_query_gives_all_nan_highest_index_first():
    from hippomm_amd.vector_ops import top_k_cosine_similarity
    store = np.random.default_rng(1).standard_normal((100, 1024), dtype=np.float32)
    idx, sims = top_k_cosine_similarity(np.zeros(1024, np.float32), store, 4)
    assert idx.tolist() == [99, 98, 97, 96] and np.isnan(sims).all()


def test_sharded_keys_merge_equals_single_scan():
    from hippomm_amd.vector_ops import FeatureStore, merge_keys_device
    rng = np.random.default_rng(21)
    store = rng.standard_normal((30000, 1024), dtype=np.float32)
    store[12345] = store[222]                              # a cross-shard tie
    q = rng.standard_normal(1024, dtype=np.float32)
    k = 32
    want_idx, want_sims = FeatureStore(store).search(q, k)
    bounds = [0, 9000, 9001, 21000, 30000]                 # ragged shards, one of a single row
    qd = torch.from_numpy(q).cuda()
    keys = torch.stack([FeatureStore(store[a:b]).search_keys_device(qd, k)
                        for a, b in zip(bounds[:-1], bounds[1:])])
    offs = torch.tensor(bounds[:-1], dtype=torch.int64, device="cuda")
    idx, sims = merge_keys_device(keys, offs, k)
    assert idx.cpu().tolist() == want_idx.tolist()
    np.testing.assert_array_equal(sims.cpu().numpy(), want_sims)


def test_full_size_properties_1m_rows():
    """BASELINE cfg 4 shape.  The oracle needs seconds per pass at this size, so check
    properties: returned sims equal an fp64 recomputation on the returned rows; order is
    non-increasing; nothing outside the result beats the k-th entry (torch fp32 matmul as an
    independent witness, compared with the scan's own tolerance); and the oracle agrees on a
    subsample that contains the winners."""
    from hippomm_amd.vector_ops import FeatureStore
    n, k = 1_000_000, 32
    g = torch.Generator(device="cuda").manual_seed(42)
    rows = torch.empty(n, 1024, dtype=torch.float32, device="cuda")
    for s in range(0, n, 125_000):
        blk = torch.randn(125_000, 1024, generator=g, device="cuda")
        rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
    q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
    store = FeatureStore(rows)
    idx, sims = store.search_device(q, k)
    idx_h, sims_h = idx.cpu().numpy(), sims.cpu().numpy()
    assert len(set(idx_h.tolist())) == k and np.all(sims_h[:-1] >= sims_h[1:])
    exact = (rows[idx].double() @ q.double()) / (rows[idx].double().norm(dim=1) * q.double().norm())
    np.testing.assert_allclose(sims_h, exact.cpu().numpy(), rtol=0, atol=SIM_ATOL)
    witness = (rows @ q) / (rows.norm(dim=1) * q.norm())
    witness[idx] = -2.0
    assert float(witness.max()) <= float(sims_h[-1]) + SIM_ATOL
    # oracle on a 20k-row subsample that includes the winners
    pick = np.unique(np.concatenate([idx_h, np.random.default_rng(0).integers(0, n, 20000)]))
    sub = rows[torch.from_numpy(pick).cuda()].cpu().numpy()
    o_idx, o_sims = top_k_cosine_similarity_oracle(q.cpu().numpy(), sub, k)
    assert pick[o_idx].tolist() == idx_h.tolist()
    np.testing.assert_allclose(sims_h, o_sims, rtol=0, atol=SIM_ATOL)


def test_similarity_pass_of_a_store_larger_than_one_epoch():
    """hmm_op_scan_sims (the similarity pass of the per-event scan and of k > 128) keeps its results in registers and writes them
    once per 512 iterations: 1.7M rows need a second epoch on the scan's grid (1536 waves x 2 rows x 512 = 1 572 864 rows per
    epoch).  Every similarity against torch within the scan's tolerance, and the rows around and beyond the epoch boundary bitwise
    equal to the same rows scanned as a store of their own (one epoch): a row's arithmetic does not depend on where it sits."""
    from hippomm_amd import _lib as L
    lib = L.load()
    n = 1_700_000 + 3                                        # odd: the last pair has one row
    g = torch.Generator(device="cuda").manual_seed(7)
    rows = torch.empty(n, 1024, dtype=torch.float32, device="cuda")
    for s in range(0, n, 100_000):
        m = min(100_000, n - s)
        rows[s:s + m] = torch.randn(m, 1024, generator=g, device="cuda")
    q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(8), device="cuda")
    sims = torch.full((n,), -7.0, device="cuda")
    L.check(lib.hmm_op_scan_sims(rows.data_ptr(), n, q.data_ptr(), sims.data_ptr(), L.stream_ptr()), "hmm_op_scan_sims")
    want = (rows @ q) / (rows.norm(dim=1) * q.norm())
    assert float((sims - want).abs().max()) <= SIM_ATOL
    lo = 1_560_000                                           # 16-byte aligned start (lo x 4096 B), before the boundary
    part = torch.empty(n - lo, device="cuda")
    L.check(lib.hmm_op_scan_sims(rows[lo:].data_ptr(), n - lo, q.data_ptr(), part.data_ptr(), L.stream_ptr()), "hmm_op_scan_sims")
    assert torch.equal(part.view(torch.int32), sims[lo:].view(torch.int32))


def test_store_cache_keeps_numpy_stores_resident_and_notices_what_it_can():
    """enable_store_cache(): the reference's unchanged per-event loop passes the same host arrays question after question; they
    are served from HBM from the second call on, rebuilt when the array object, its buffer or a sampled element changed, released
    when the host array dies, and the feature is off unless asked for."""
    import gc
    from hippomm_amd import vector_ops as vo
    rng = np.random.default_rng(31)
    events = [rng.standard_normal((n, 1024)).astype(np.float64 if i % 2 else np.float32) for i, n in enumerate((300, 41, 1000))]
    q = rng.standard_normal(1024, dtype=np.float32)
    want = [vo.top_k_cosine_similarity(q, ev, 5) for ev in events]
    assert vo._STORE_CACHE is None                                          # off by default
    cache = vo.enable_store_cache(max_bytes=64 << 20)
    try:
        for rnd in range(3):
            for ev, (wi, ws) in zip(events, want):
                idx, sims = vo.top_k_cosine_similarity(q, ev, 5)
                assert idx.tolist() == wi.tolist() and sims.dtype == ws.dtype and np.array_equal(sims, ws)
        assert (cache.misses, cache.hits) == (3, 6)
        events[0][0, 0] += 1.0                                              # row 0 is always in the fingerprint sample
        idx, sims = vo.top_k_cosine_similarity(q, events[0], 5)
        with np.errstate(invalid="ignore"):
            o_idx, o_sims = top_k_cosine_similarity_oracle(q, events[0], 5)
        assert cache.misses == 4 and idx.tolist() == o_idx.tolist()
        held = cache.bytes
        del events[2], ev                                                   # `ev` still names the last event of the loop above
        gc.collect()
        assert cache.bytes == held - 1000 * 4096 and len(cache.entries) == 2   # the HBM copy went with the host array
        big = rng.standard_normal((20000, 1024), dtype=np.float32)          # 80 MB > max_bytes: evicts the others, still answers
        vo.top_k_cosine_similarity(q, big, 5)
        assert len(cache.entries) == 1
    finally:
        vo.disable_store_cache()
    assert vo._STORE_CACHE is None
