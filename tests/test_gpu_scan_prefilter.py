"""GPU parity of the bf16-prefilter scan (hmm_cosine_topk_prefilter, SURVEY 8d's optional shadow store): it must return what
hmm_cosine_topk returns -- the same rows and the same fp32 similarities BIT FOR BIT -- on random stores, on stores where the
bf16 rounding error is larger than the gaps between ranks (clusters of near-duplicate rows, as consecutive video frames are),
on ties, NaN rows and a zero query, at sizes on both sides of its dispatch limits, on the reference's golden vectors, and at the
BASELINE size.  The reference function is hippomm/utils/vector_ops.py:151-188."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import recipes
from test_gpu_scan import GOLD, SIM_ATOL, _sims, assert_topk_matches

pytestmark = pytest.mark.gpu


def both(store, q, k):
    """(exact idx, exact sims, prefiltered idx, prefiltered sims, stats) as numpy."""
    from hippomm_amd.vector_ops import FeatureStore
    fs = FeatureStore(store)
    qd = q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).cuda()
    stats = torch.full((2,), -7, dtype=torch.int32, device="cuda")
    i0, s0 = fs.search_device(qd, k)
    i1, s1 = fs.search_prefiltered_device(qd, k, stats)
    return i0.cpu().numpy(), s0.cpu().numpy(), i1.cpu().numpy(), s1.cpu().numpy(), stats.cpu().tolist()


def assert_identical(i0, s0, i1, s1):
    assert i1.dtype == np.int64 and i0.tolist() == i1.tolist()
    assert s0.view(np.int32).tolist() == s1.view(np.int32).tolist()      # the same bits, NaN included


@pytest.mark.parametrize("n,k", [(1, 1), (63, 5), (4097, 32), (16383, 32), (16384, 32), (16385, 1), (20000, 5), (20000, 64),
                                 (20000, 65), (70001, 32), (300001, 64), (300001, 100)])
def test_equals_the_exact_scan_on_random_stores(n, k):
    rng = np.random.default_rng(n * 17 + k)
    store = rng.standard_normal((n, 1024), dtype=np.float32)
    q = rng.standard_normal(1024, dtype=np.float32)
    i0, s0, i1, s1, stats = both(store, q, k)
    assert_identical(i0, s0, i1, s1)
    assert_topk_matches(i1, s1, store, q, k)
    if n >= 16384 and k <= 64:
        assert stats[1] == 0 and k <= stats[0] <= 4 * k + 64, stats        # the prefilter answered, with few extra candidates
    else:
        assert stats == [-1, -1]                                             # below the dispatch limits the call IS the exact scan


def test_clusters_of_near_duplicates_where_bf16_error_exceeds_the_rank_gaps():
    """Scenes of 40 near-identical frames: inside a scene the similarities to a query differ by ~1e-5 while the shadow's error is
    ~1e-3, so the approximate ranking inside the winning scenes is wrong and only the margin + exact re-score can be right."""
    rng = np.random.default_rng(5)
    scenes = rng.standard_normal((600, 1024), dtype=np.float32)
    store = np.repeat(scenes, 40, axis=0) + 2e-3 * rng.standard_normal((24000, 1024), dtype=np.float32)
    q = scenes[123] + 0.5 * scenes[77] + 0.05 * rng.standard_normal(1024, dtype=np.float32)
    for k in (5, 32, 64):
        i0, s0, i1, s1, stats = both(store, q, k)
        assert_identical(i0, s0, i1, s1)
        assert stats[0] >= 40 or stats[1] > 0, stats                         # the whole winning scene had to be re-scored
    # the approximate order really is wrong here: the bf16 similarities of the winning scene rank differently
    shadow = torch.from_numpy(store[123 * 40:124 * 40]).cuda()
    shadow = (shadow / shadow.norm(dim=1, keepdim=True)).to(torch.bfloat16).float()
    approx = (shadow @ torch.from_numpy(q).cuda()).cpu().numpy()
    exact = (store[123 * 40:124 * 40].astype(np.float64) @ q.astype(np.float64)) / np.linalg.norm(store[123 * 40:124 * 40].astype(np.float64), axis=1)
    assert np.argsort(-approx).tolist() != np.argsort(-exact).tolist()


def test_query_aligned_with_the_rounding_error_of_the_best_row():
    """The query leans against the bf16 rounding error of its own best match, so that row's approximate similarity is pushed
    down by ~1e-3 -- below decoys whose true similarity is lower.  The exact answer must still come back."""
    rng = np.random.default_rng(9)
    store = rng.standard_normal((30000, 1024), dtype=np.float32)
    v = store[4321] / np.linalg.norm(store[4321])
    vt = torch.from_numpy(v).to(torch.bfloat16).float().numpy()
    e = vt - v
    q = (v - 0.6 * e / np.linalg.norm(e)).astype(np.float32)
    for j in range(64):                                                       # decoys: a little further from v than v itself
        store[100 + 7 * j] = v + (0.02 + 0.0005 * j) * rng.standard_normal(1024).astype(np.float32) / 32
    i0, s0, i1, s1, stats = both(store, q, 32)
    assert_identical(i0, s0, i1, s1)
    assert 4321 in i1.tolist()


def test_many_exact_ties_and_nan_rows_take_the_fallback_and_stay_identical():
    rng = np.random.default_rng(11)
    base = rng.standard_normal((16, 1024), dtype=np.float32)
    store = base[rng.integers(0, 16, size=40000)]                             # every row ~2500 times
    store[[17, 4000, 39999]] = 0.0                                            # NaN rows rank first
    q = rng.standard_normal(1024, dtype=np.float32)
    for k in (3, 32):
        i0, s0, i1, s1, stats = both(store, q, k)
        assert_identical(i0, s0, i1, s1)
        assert i1[:3].tolist() == [39999, 4000, 17] and np.isnan(s1[:3]).all()
    assert stats[1] > 0 or stats[0] > 1024, stats                             # 2500 equal best rows: more candidates than pass 2 re-scores, the exact scan answered


def test_zero_and_nan_queries():
    store = np.random.default_rng(1).standard_normal((20000, 1024), dtype=np.float32)
    for q in (np.zeros(1024, np.float32), np.full(1024, np.nan, np.float32)):
        i0, s0, i1, s1, _ = both(store, q, 4)
        assert_identical(i0, s0, i1, s1)
        assert i1.tolist() == [19999, 19998, 19997, 19996] and np.isnan(s1).all()


def test_text_query_shaped_case():
    """A cross-modal query: norm 1 / 0.07, similarities 0.04 - 0.4 against unit rows (the gaps of a real text -> vision query)."""
    rng = np.random.default_rng(21)
    rows = rng.standard_normal((50000, 1024), dtype=np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    q = (rows[777] * 0.3 + rng.standard_normal(1024).astype(np.float32) / 32)
    q = (q / np.linalg.norm(q) / 0.07).astype(np.float32)
    i0, s0, i1, s1, stats = both(rows, q, 5)
    assert_identical(i0, s0, i1, s1)
    assert i1[0] == 777 and stats[1] == 0


@pytest.mark.parametrize("name", recipes.SCAN_CASES)
def test_golden_vectors_through_the_prefilter_entry_point(name):
    from hippomm_amd.vector_ops import FeatureStore
    case = GOLD[name]
    q, store, k = recipes.scan_case(name)
    if k < 1 or len(np.atleast_2d(store)) == 0:
        pytest.skip("k <= 0 / empty store are host-side conventions of top_k_cosine_similarity")
    fs = FeatureStore(store)
    qd = torch.from_numpy(np.ascontiguousarray(np.asarray(q).reshape(-1), dtype=np.float32)).cuda()
    idx, sims = fs.search_prefiltered_device(qd, k)
    assert idx.cpu().tolist() == case["indices"]
    np.testing.assert_allclose(sims.cpu().numpy(), _sims(case), rtol=0, atol=SIM_ATOL, equal_nan=True)


def test_full_size_1m_rows_identical_to_the_exact_scan():
    from hippomm_amd.vector_ops import FeatureStore
    n, k = 1_000_000, 32
    g = torch.Generator(device="cuda").manual_seed(42)
    rows = torch.empty(n, 1024, dtype=torch.float32, device="cuda")
    for s in range(0, n, 125_000):
        blk = torch.randn(125_000, 1024, generator=g, device="cuda")
        rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
    fs = FeatureStore(rows).build_shadow()
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    for seed in (43, 44, 45):
        q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(seed), device="cuda")
        i0, s0 = fs.search_device(q, k)
        i1, s1 = fs.search_prefiltered_device(q, k, stats)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
        st = stats.cpu().tolist()
        assert st[1] == 0 and k <= st[0] <= 200, st
    # the shadow is what the header says: bf16(row / ||row||), row-major
    sh = fs._shadow.view(torch.bfloat16).view(n, 1024)
    pick = torch.tensor([0, 1, 499_999, 999_999], device="cuda")
    want = (rows[pick] / rows[pick].norm(dim=1, keepdim=True)).to(torch.bfloat16)
    assert (sh[pick].float() - want.float()).abs().max().item() <= 2 ** -8 * want.float().abs().max().item()


def _events(sizes, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((n, 1024), dtype=np.float32) for n in sizes]


@pytest.mark.parametrize("sizes,k", [([300, 1, 0, 57, 5, 4096, 4097, 2], 5), ([9000, 3, 12000], 32), ([40] * 200, 5), ([500] * 64, 5),
                                     ([1500, 2, 0, 700, 1025, 64, 3000, 1], 5), ([200] * 50, 64),
                                     ([5000], 64), ([5000], 100)])
def test_per_event_prefilter_equals_the_exact_per_event_scan(sizes, k):
    """hmm_cosine_topk_segmented_prefilter vs hmm_cosine_topk_segmented: identical indices, similarity bits and counts for every
    event -- empty and one-row events, events above one chunk, a tie and a NaN row inside an event, k above the prefilter's limit."""
    from hippomm_amd.vector_ops import EventStore
    events = _events(sizes, seed=len(sizes) * 7 + k)
    if len(events) > 3 and events[3].shape[0] > 10:
        events[3][7] = events[3][2]
        events[3][9] = 0.0
    es = EventStore(events)
    q = torch.from_numpy(np.random.default_rng(1).standard_normal(1024, dtype=np.float32)).cuda()
    i0, s0, c0 = es.search_segments_device(q, es.offsets, k)
    i1, s1, c1 = es.search_segments_device(q, es.offsets, k, prefilter=True)
    assert torch.equal(c0, c1) and torch.equal(i0, i1)
    assert torch.equal(s0.view(torch.int32), s1.view(torch.int32))


def test_per_event_prefilter_on_a_store_larger_than_one_epoch_of_its_similarity_pass():
    """prefilter_sims_deferred_kernel writes its results once per 256 iterations: 768 workgroups x 4 waves x 4 rows x 256 =
    3 145 728 rows per epoch.  3.2M rows in events of 500: the per-event answer through the shadow equals the exact one (indices,
    similarity bits, counts) on both sides of the boundary and in the ragged last event."""
    from hippomm_amd.vector_ops import EventStore
    n = 3_200_000 + 7
    g = torch.Generator(device="cuda").manual_seed(11)
    rows = torch.empty(n, 1024, dtype=torch.float32, device="cuda")
    for s in range(0, n, 100_000):
        m = min(100_000, n - s)
        rows[s:s + m] = torch.randn(m, 1024, generator=g, device="cuda")
    sizes = [500] * (n // 500) + [n % 500]
    es = EventStore.from_device_rows(rows, sizes)
    q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(12), device="cuda")
    i0, s0, c0 = es.search_segments_device(q, es.offsets, 5)
    i1, s1, c1 = es.search_segments_device(q, es.offsets, 5, prefilter=True)
    assert torch.equal(c0, c1) and torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    e = 3_145_728 // 500                                      # the event that straddles the epoch boundary: against torch
    lo = e * 500
    ev = rows[lo:lo + 500]
    want = torch.topk((ev @ q) / (ev.norm(dim=1) * q.norm()), 5)
    assert i1[e].tolist() == want.indices.tolist()


def test_per_event_prefilter_on_events_of_near_identical_rows():
    """Every event is one scene: 300 frames within 1e-3 of each other, so every row of an event is a candidate (more than the
    candidate buffer for the large event) and the whole event is re-scored; the answer must still be the exact one."""
    from hippomm_amd.vector_ops import EventStore
    rng = np.random.default_rng(3)
    sizes = [300, 2000, 40, 300]
    events = []
    for n in sizes:
        c = rng.standard_normal(1024).astype(np.float32)
        events.append(c + 1e-3 * rng.standard_normal((n, 1024), dtype=np.float32))
    es = EventStore(events)
    q = torch.from_numpy((events[1][5] + 0.3 * rng.standard_normal(1024)).astype(np.float32)).cuda()
    for k in (5, 32):
        i0, s0, c0 = es.search_segments_device(q, es.offsets, k)
        i1, s1, c1 = es.search_segments_device(q, es.offsets, k, prefilter=True)
        assert torch.equal(c0, c1) and torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    hits0, hits1 = es.top_hits(q, 5, 5), es.top_hits(q, 5, 5, prefilter=True)
    assert hits0 == hits1


def test_error_codes_through_the_c_abi():
    """Bad arguments come back as status codes with a message (include/hippomm_hip.h conventions), never as a crash."""
    import ctypes as C
    from hippomm_amd import _lib as L
    lib = L.load()
    n, k = 20000, 5
    rows = torch.randn(n, 1024, device="cuda")
    q = torch.randn(1024, device="cuda")
    shadow = torch.empty(lib.hmm_shadow_store_bytes(n), dtype=torch.uint8, device="cuda")
    assert lib.hmm_shadow_store_bytes(n) == n * 2048 and lib.hmm_shadow_store_bytes(0) == 0
    idx = torch.empty(k, dtype=torch.int64, device="cuda"); sims = torch.empty(k, device="cuda")
    n_out = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.hmm_cosine_topk_prefilter_workspace_bytes(n, k), dtype=torch.uint8, device="cuda")
    st = L.stream_ptr()
    assert lib.hmm_shadow_store_build(rows.data_ptr(), n, 1024, shadow.data_ptr(), shadow.numel(), st) == 0
    assert lib.hmm_shadow_store_build(rows.data_ptr(), n, 512, shadow.data_ptr(), shadow.numel(), st) == -1          # HMM_E_INVALID
    assert lib.hmm_shadow_store_build(rows.data_ptr(), n, 1024, shadow.data_ptr(), shadow.numel() - 1, st) == -2     # HMM_E_WORKSPACE
    assert b"shadow buffer" in lib.hmm_last_error()
    call = lambda **kw: lib.hmm_cosine_topk_prefilter(kw.get("store", rows.data_ptr()), kw.get("shadow", shadow.data_ptr()), n, 1024,
                                                      q.data_ptr(), kw.get("k", k), idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(),
                                                      None, ws.data_ptr(), kw.get("ws_bytes", ws.numel()), st)
    assert call() == 0 and int(n_out.item()) == k
    assert call(ws_bytes=1024) == -2
    assert call(k=0) == -1 and call(shadow=None) == -1
    assert call(store=rows.data_ptr() + 4) == -1 and b"aligned" in lib.hmm_last_error()
    with pytest.raises(L.HippoMMHipError):
        L.check(call(k=-3), "hmm_cosine_topk_prefilter")
    # the per-event entry point: no segments / k = 0 on a non-empty store are argument errors, not a host division by zero
    offs = torch.tensor([0, n], dtype=torch.int64, device="cuda")
    seg_ws = torch.empty(max(lib.hmm_cosine_topk_segmented_prefilter_workspace_bytes(n, 1, k), 1024), dtype=torch.uint8, device="cuda")
    seg = lambda n_seg, kk: lib.hmm_cosine_topk_segmented_prefilter(rows.data_ptr(), shadow.data_ptr(), n, 1024, q.data_ptr(), offs.data_ptr(),
                                                                     n_seg, kk, idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(),
                                                                     seg_ws.data_ptr(), seg_ws.numel(), st)
    assert seg(1, k) == 0 and int(n_out.item()) == k
    assert seg(0, k) == -1 and b"n_segments" in lib.hmm_last_error()
    assert seg(-2, k) == -1
    assert seg(1, 0) == -1
    # a workspace that lost its 16-byte alignment is refused (include/hippomm_hip.h, Alignment), here and by the exact entry points
    big = torch.empty(seg_ws.numel() + max(lib.hmm_cosine_topk_segmented_workspace_bytes(n, 1, k), lib.hmm_cosine_topk_workspace_bytes(n, k)) + 64,
                      dtype=torch.uint8, device="cuda")
    assert lib.hmm_cosine_topk_segmented_prefilter(rows.data_ptr(), shadow.data_ptr(), n, 1024, q.data_ptr(), offs.data_ptr(), 1, k, idx.data_ptr(),
                                                   sims.data_ptr(), n_out.data_ptr(), big.data_ptr() + 4, big.numel() - 4, st) == -1
    assert b"aligned" in lib.hmm_last_error()
    assert lib.hmm_cosine_topk_segmented(rows.data_ptr(), n, 1024, q.data_ptr(), offs.data_ptr(), 1, k, idx.data_ptr(), sims.data_ptr(),
                                         n_out.data_ptr(), big.data_ptr() + 8, big.numel() - 8, st) == -1
    assert lib.hmm_cosine_topk(rows.data_ptr(), n, 1024, q.data_ptr(), k, idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(),
                               big.data_ptr() + 4, big.numel() - 4, st) == -1
    assert lib.hmm_cosine_topk(rows.data_ptr(), n, 1024, q.data_ptr(), k, idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(),
                               big.data_ptr() + 16, big.numel() - 16, st) == 0


def test_feature_store_with_shadow_serves_the_drop_in_call_identically():
    """FeatureStore(rows, shadow=True): top_k_cosine_similarity(q, store, k) goes through the shadow and returns what it returns
    without it (indices, similarity bits, dtypes), for float32 and float64 sources and k on both sides of the prefilter's limit."""
    from hippomm_amd.vector_ops import FeatureStore, top_k_cosine_similarity
    rng = np.random.default_rng(77)
    for dtype in (np.float32, np.float64):
        store = rng.standard_normal((30000, 1024)).astype(np.float32).astype(dtype)
        q = rng.standard_normal(1024).astype(np.float32)
        plain, shadowed = FeatureStore(store), FeatureStore(store, shadow=True)
        assert shadowed.use_shadow and not plain.use_shadow
        for k in (5, 32, 100):
            i0, s0 = top_k_cosine_similarity(q, plain, k)
            i1, s1 = top_k_cosine_similarity(q, shadowed, k)
            assert i0.tolist() == i1.tolist() and s0.dtype == s1.dtype and s0.tobytes() == s1.tobytes()


def test_shadow_follows_in_place_updates_of_the_rows():
    """The bf16 shadow is a snapshot of the rows (round-4 advisor finding: a stale shadow silently drops true top-k rows).  An
    in-place update through torch moves the tensor's version counter and the next prefiltered search rebuilds the shadow;
    `invalidate_shadow()` / `build_shadow(force=True)` cover writes torch does not see."""
    from hippomm_amd.vector_ops import FeatureStore
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.randn(40000, 1024, generator=g, device="cuda")
    q = torch.randn(1024, generator=g, device="cuda")
    store = FeatureStore(rows, shadow=True)                    # aliases `rows` (fp32, contiguous, already on the device)
    assert store.rows.data_ptr() == rows.data_ptr()
    i0, s0 = store.search_device(q, 8)
    # plant a perfect match in a row that was nowhere near the top: the exact scan and the shadow path must both find it
    target = int((torch.arange(40000, device="cuda")[~torch.isin(torch.arange(40000, device="cuda"), i0)])[12345].item())
    rows[target] = q * 3.0                                     # in place, through torch: version counter moves
    i1, s1 = store.search_device(q, 8)                         # use_shadow=True -> the prefilter path, shadow rebuilt first
    exact = FeatureStore(rows)
    ie, se = exact.search_device(q, 8)
    assert int(i1[0].item()) == target and torch.equal(i1, ie) and torch.equal(s1.view(torch.int32), se.view(torch.int32))
    before = store._shadow.clone()
    store.build_shadow()                                       # nothing changed since: no rebuild
    assert torch.equal(store._shadow, before)
    store.build_shadow(force=True)
    assert torch.equal(store._shadow, before)                  # same rows -> same shadow bits
    store.invalidate_shadow()
    assert store._shadow is None
    i2, s2 = store.search_device(q, 8)
    assert torch.equal(i2, ie) and torch.equal(s2.view(torch.int32), se.view(torch.int32))


def test_shadow_of_a_store_created_under_inference_mode():
    """An inference tensor tracks no version counter (reading ._version raises RuntimeError, not AttributeError): such a store is a
    snapshot until invalidate_shadow(), and the prefiltered searches work on it."""
    from hippomm_amd.vector_ops import FeatureStore
    with torch.inference_mode():
        g = torch.Generator(device="cuda").manual_seed(3)
        rows = torch.randn(20000, 1024, generator=g, device="cuda")
        q = torch.randn(1024, generator=g, device="cuda")
        store = FeatureStore(rows)
        store.build_shadow()
        i0, s0 = store.search_device(q, 5)
        i1, s1 = store.search_prefiltered_device(q, 5)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
