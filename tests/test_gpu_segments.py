"""GPU parity of the per-event scan: one launch over all events must return, for every event,
exactly what the reference's per-event call top_k_cosine_similarity(q, event_features, k) returns."""
import numpy as np
import pytest

from oracle.vector_ops_oracle import top_k_cosine_similarity_oracle

pytestmark = pytest.mark.gpu
SIM_ATOL = 2e-6


def _events(sizes, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((n, 1024), dtype=np.float32) for n in sizes]


@pytest.mark.parametrize("sizes,k", [([300, 1, 0, 57, 5, 4096, 4097, 2], 5),       # reference's k=5; empty / tiny events
                                     ([9000, 3, 12000], 32),                        # events larger than one chunk
                                     ([40] * 200, 5),                               # many small events
                                     ([1500, 2, 0, 700, 1025, 64, 3000, 1], 5),     # small mean (the 1024-key kernels), events above it
                                     ([200] * 50, 64),                              # ... at their largest k
                                     ([5000], 1024)])
def test_per_event_matches_reference_per_event_calls(sizes, k):
    from hippomm_amd.vector_ops import EventStore
    events = _events(sizes, seed=len(sizes) * 7 + k)
    if len(events) > 3 and events[3].shape[0] > 10:
        events[3][7] = events[3][2]                       # a tie inside an event
        events[3][9] = 0.0                                # a NaN row
    q = np.random.default_rng(1).standard_normal(1024, dtype=np.float32)
    got = EventStore(events).top_k_per_event(q, k)
    assert len(got) == len(events)
    for ev, (idx, sims) in zip(events, got):
        if ev.shape[0] == 0:
            assert len(idx) == 0 and len(sims) == 0
            continue
        with np.errstate(invalid="ignore", divide="ignore"):
            want_idx, want_sims = top_k_cosine_similarity_oracle(q, ev, k)
        assert idx.dtype == np.int64 and len(idx) == min(k, ev.shape[0])
        np.testing.assert_allclose(sims, want_sims, rtol=0, atol=SIM_ATOL, equal_nan=True)
        if k <= 32:                                       # well separated: indices exact
            assert idx.tolist() == want_idx.tolist()
        else:
            with np.errstate(invalid="ignore", divide="ignore"):
                all_sims = (ev @ q) / (np.linalg.norm(ev, axis=1) * np.linalg.norm(q))
            np.testing.assert_allclose(all_sims[idx], want_sims, rtol=0, atol=SIM_ATOL)


def test_segmented_equals_single_scan_on_one_segment():
    import torch
    from hippomm_amd.vector_ops import FeatureStore
    store = np.random.default_rng(3).standard_normal((20000, 1024), dtype=np.float32)
    q = torch.from_numpy(np.random.default_rng(4).standard_normal(1024, dtype=np.float32)).cuda()
    fs = FeatureStore(store)
    idx, sims = fs.search_device(q, 32)
    off = torch.tensor([0, 20000], dtype=torch.int64, device="cuda")
    sidx, ssims, cnt = fs.search_segments_device(q, off, 32)
    assert int(cnt[0]) == 32 and torch.equal(sidx[0], idx) and torch.equal(ssims[0], sims)


def test_build_event_store_from_memory_store_dir(tmp_path):
    """memory_store/ on disk (reference layout) -> EventStore in HBM -> per-event top-5 == oracle per event."""
    import json
    import recipes
    from hippomm_amd import event_store as es
    base = tmp_path / "memory_store"
    rng = np.random.default_rng(12)
    index, mats = {}, []
    for i, n in enumerate([7, 300, 1, 50]):
        case = recipes.event_case()
        v = rng.standard_normal((n, 1024)).astype(np.float32)
        case["features"] = {"vision": v, "vision_times": np.arange(n, dtype=np.float64)}
        if i == 2:
            case["features"] = {"audio": v, "audio_times": np.arange(n, dtype=np.float64)}     # no vision in this event
        eid = f"vid_{i * 1000}"
        p = es.save_event(case, base / "events" / "vid" / f"{eid}.json", write_sidecars=(i % 2 == 0))
        index[eid] = {"video_id": "vid", "start_time": float(i), "end_time": float(i + 1), "file_path": str(p)}
        mats.append(v if i != 2 else np.zeros((0, 1024), np.float32))
    (base / "event_index.json").write_text(json.dumps(index, indent=2))
    store, ids = es.build_event_store(base, "vision")
    assert ids == list(index) and store.lengths == [7, 300, 0, 50]
    q = rng.standard_normal(1024).astype(np.float32)
    for ev, (idx, sims) in zip(mats, store.top_k_per_event(q, 5)):
        if ev.shape[0] == 0:
            assert len(idx) == 0
            continue
        want_idx, want_sims = top_k_cosine_similarity_oracle(q, ev, 5)
        assert idx.tolist() == want_idx.tolist()
        np.testing.assert_allclose(sims, want_sims, rtol=0, atol=SIM_ATOL)
