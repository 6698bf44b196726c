"""GPU: the retrieval path end to end (SURVEY 8f-1 + 8f-2 + 8f-4 together), as QARecallSystem runs it:

    question string -> tokenizer -> text tower -> query (1024,)            hippocampal_memory.py:2173-2176
    for every event: top_k_cosine_similarity(query, event.features, k=5)   hippocampal_memory.py:3143-3153
    all hits sorted by similarity, best 5 kept                             hippocampal_memory.py:3275-3277

here: SimpleTokenizer -> ImageBind.extract_features({'text': [...]}) -> build_event_store(memory_store dir) ->
EventStore.top_k_per_event(k=5), against oracle/text_forward for the embedding and the scan oracle called per event on the
matrices the reference's own reader yields (JSON -> float64).  Two stores: a memory_store DIRECTORY of 200 events written
through save_event (JSON byte-identical to the reference's + sidecars; ~5 k rows: the JSON costs 29 bytes per stored value)
and an in-memory EventStore of 250 events / 120 k rows for the same query."""
import json

import numpy as np
import pytest
import torch

from oracle import imagebind_oracle as ib
from oracle.vector_ops_oracle import top_k_cosine_similarity_oracle

pytestmark = pytest.mark.gpu
QUESTION = "who opens the door after the phone rang?"


def _ranked(per_event, keep=5):
    """The caller's final step (:3261-3277): every hit of every event, best similarity first, keep 5."""
    hits = [(float(s), e, int(i)) for e, (idx, sims) in enumerate(per_event) for i, s in zip(idx, sims)]
    hits.sort(key=lambda h: -h[0])
    return hits[:keep]


def _same_rows_where_separated(q, rows, idx, sims, w_idx, band=4e-6):
    """Every returned row carries the similarity reported for it; where a rank is clear of BOTH its neighbours in the full
    ranking (the next-best row that was not returned included) it is the reference's row.  Inside a tie group -- the
    duplicated frame -- the reference's order is numpy's unstable argsort, the library's is 'higher row first'."""
    if len(idx) == 0:
        return
    with np.errstate(invalid="ignore", divide="ignore"):
        all_sims = (rows @ q) / (np.linalg.norm(rows, axis=1) * np.linalg.norm(q))
    np.testing.assert_allclose(all_sims[idx], sims, rtol=0, atol=2e-6)
    order = np.sort(all_sims)[::-1]
    m = len(idx)
    gaps = np.full(m + 1, np.inf)
    top = order[: m + 1]
    gaps[1: len(top)] = top[:-1] - top[1:]
    separated = (gaps[:m] > band) & (gaps[1: m + 1] > band)
    assert np.array_equal(np.asarray(idx)[separated], np.asarray(w_idx, dtype=np.int64)[separated])


@pytest.fixture(scope="module")
def text_model():
    from test_tokenizer import _learn_merges
    from hippomm_amd.encoder import ImageBind, synthetic_state_dict
    from hippomm_amd.tokenizer import SimpleTokenizer
    tok = SimpleTokenizer("", merges=_learn_merges())
    sd = synthetic_state_dict(("text",), seed=4321)
    model = ImageBind(state_dict=sd, towers=("text",), tokenizer=tok)
    st = {k: v.detach().float().cpu() for k, v in sd.items()}
    return model, tok, st


@pytest.fixture(scope="module")
def query(text_model):
    model, tok, st = text_model
    emb = model.extract_features({"text": [QUESTION]}, ["text"])["text"]                 # (1, 1024) on the GPU
    q = emb.cpu().numpy().flatten()                                                      # as :3130-3134
    tokens = tok([QUESTION])
    assert tokens.shape == (1, 77) and int(tokens[0].argmax()) == len(tok.encode(QUESTION)) + 1
    want = ib.text_forward(tokens, st).numpy().flatten()                                 # fp32 oracle, all 24 blocks
    cos = float(np.dot(q, want) / (np.linalg.norm(q) * np.linalg.norm(want)))
    assert 1 - cos <= 5e-5, cos
    assert abs(np.linalg.norm(q) - np.linalg.norm(want)) <= 1e-3 * np.linalg.norm(want)  # x exp(log_logit_scale)
    return q


def test_question_to_hits_over_a_memory_store_directory(tmp_path, query):
    from hippomm_amd import event_store as es
    rng = np.random.default_rng(11)
    base = tmp_path / "memory_store"
    index, sizes = {}, []
    for e in range(200):
        n = int(rng.integers(1, 48)) if e % 17 else 0                                    # a few events without frames
        f = rng.standard_normal((n, 1024)).astype(np.float32)
        if n > 6:
            f[5] = f[2]                                                                   # a duplicated frame (tie)
        vid = f"video_{e // 10:02d}"
        event_id = f"{vid}_{e * 30000}"
        path = base / "events" / vid / f"{event_id}.json"
        es.save_event({"features": {"vision": f, "vision_times": np.arange(n) * 1.0}, "frames": [], "frame_times": [],
                       "frame_captions": [], "audio_times": [], "audio_transcription": [], "holistic_audio_transcription": "",
                       "summary": f"event {e}", "start_time": e * 30.0, "end_time": e * 30.0 + 29.0}, path)
        index[event_id] = {"video_id": vid, "start_time": e * 30.0, "end_time": e * 30.0 + 29.0, "file_path": str(path)}
        sizes.append(n)
    (base / "event_index.json").write_text(json.dumps(index, indent=2))
    store, ids = es.build_event_store(base, "vision")
    assert ids == list(index) and store.lengths == sizes and sum(sizes) > 4000
    got = store.top_k_per_event(query, 5)
    want = []
    for event_id, path in es.iter_event_files(base):
        data = json.loads(path.read_text())                                              # the reference's reader: json.load +
        rows = np.array(data["features"]["vision"])                                      # np.array(list) -> float64 (:387-395)
        if rows.size == 0:
            want.append((np.zeros(0, np.int64), np.zeros(0)))
            continue
        assert rows.dtype == np.float64
        want.append(top_k_cosine_similarity_oracle(query, rows, 5))
    assert len(got) == len(want) == 200
    mats = [np.array(json.loads(path.read_text())["features"]["vision"]).reshape(-1, 1024) for _, path in es.iter_event_files(base)]
    for (idx, sims), (w_idx, w_sims), n, rows in zip(got, want, sizes, mats):
        assert idx.dtype == np.int64 and len(idx) == min(5, n)
        np.testing.assert_allclose(sims, w_sims, rtol=0, atol=2e-6)
        _same_rows_where_separated(query, rows, idx, sims, w_idx)
    top_got, top_want = _ranked(got), _ranked(want)
    np.testing.assert_allclose([s for s, _, _ in top_got], [s for s, _, _ in top_want], rtol=0, atol=2e-6)
    assert [e for _, e, _ in top_got] == [e for _, e, _ in top_want]                     # a tie never crosses events here
    for (s, e, i), (_, _, wi) in zip(top_got, top_want):
        assert i == wi or np.array_equal(mats[e][i], mats[e][wi])                        # the duplicated frame


def test_question_to_hits_over_250_events_and_120k_rows(query):
    from hippomm_amd.vector_ops import EventStore
    rng = np.random.default_rng(12)
    sizes = [int(n) for n in rng.integers(200, 760, size=250)]
    events = [rng.standard_normal((n, 1024), dtype=np.float32) for n in sizes]
    assert sum(sizes) >= 100_000
    got = EventStore(events).top_k_per_event(query, 5)
    want = [top_k_cosine_similarity_oracle(query, ev, 5) for ev in events]               # the reference's loop, event by event
    for (idx, sims), (w_idx, w_sims), ev in zip(got, want, events):
        np.testing.assert_allclose(sims, w_sims, rtol=0, atol=2e-6)
        _same_rows_where_separated(query, ev.astype(np.float64), idx, sims, w_idx)
    assert [(e, i) for _, e, i in _ranked(got)] == [(e, i) for _, e, i in _ranked(want)]
    # the same final step on the device: only the five hits come back
    store = EventStore(events)
    hits = store.top_hits(query, k=5, keep=5)
    assert [(e, i) for e, i, _ in hits] == [(e, i) for _, e, i in _ranked(want)]
    np.testing.assert_allclose([s for _, _, s in hits], [s for s, _, _ in _ranked(want)], rtol=0, atol=2e-6)
    assert len(store.top_hits(query, k=5, keep=100000)) == sum(min(5, n) for n in sizes)       # keep > all hits: every hit, ranked
    few = EventStore([events[0][:2], np.zeros((0, 1024), np.float32)])
    assert [(e, i) for e, i, _ in few.top_hits(query, 5, 5)] == [(0, int(i)) for i in top_k_cosine_similarity_oracle(query, events[0][:2], 5)[0]]


def test_rank_segment_hits_is_pythons_stable_descending_sort():
    """hmm_rank_segment_hits on hand-made (E, k) outputs of the per-event scan against the reference's rule -- every event's hits in one
    list, `sorted(..., key=sim, reverse=True)` (stable: equal similarities stay in event order, then in rank order), the best kept:
    ties inside and across events, -0.0 == +0.0, events with fewer than k hits, keep beyond the number of hits, NaN first, and more
    keys than one tournament window (the quick route up to 65 536 keys, the windowed pass beyond)."""
    import ctypes as C
    import torch
    from hippomm_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(31)

    def run(sims, counts, keep):
        E, k = sims.shape
        idx = torch.from_numpy(rng.integers(0, 10_000, (E, k))).cuda()
        s, c = torch.from_numpy(sims).cuda(), torch.from_numpy(counts.astype(np.int32)).cuda()
        ev, row = torch.empty(keep, dtype=torch.int64, device="cuda"), torch.empty(keep, dtype=torch.int64, device="cuda")
        val, n = torch.empty(keep, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        L.check(lib.hmm_rank_segment_hits(idx.data_ptr(), s.data_ptr(), c.data_ptr(), E, k, keep, ev.data_ptr(), row.data_ptr(), val.data_ptr(),
                                          n.data_ptr(), None), "hmm_rank_segment_hits")
        n = int(n.item())
        flat = [(e, j) for e in range(E) for j in range(int(counts[e]))]
        nan_first = [p for p in flat if np.isnan(sims[p])] + [p for p in flat if not np.isnan(sims[p])]
        want = sorted(nan_first, key=lambda p: (0 if np.isnan(sims[p]) else 1, -float(sims[p]) if not np.isnan(sims[p]) else 0.0))[:keep]
        assert n == len(want), (n, len(want))
        got = list(zip(ev.cpu().tolist()[:n], row.cpu().tolist()[:n], val.cpu().numpy()[:n]))
        idx_h = idx.cpu().numpy()
        for (e, r, v), (we, wj) in zip(got, want):
            assert e == we and r == idx_h[we, wj] and (v == sims[we, wj] or (np.isnan(v) and np.isnan(sims[we, wj]))), ((e, r, v), (we, wj))
        assert ev.cpu().tolist()[n:] == [-1] * (keep - n)

    k = 5
    sims = rng.standard_normal((300, k)).astype(np.float32)
    sims[:, :] = -np.sort(-sims, axis=1)                                   # per event descending, as the scan leaves them
    sims[17, 0] = sims[3, 0] = sims[250, 0] = 9.0                          # a three-way tie across events: event order
    sims[40, 1] = sims[40, 0]                                              # ... and inside an event: rank order
    sims[5, 0], sims[6, 0] = 0.0, -0.0
    counts = np.full(300, k)
    counts[10], counts[11], counts[299] = 0, 2, 1
    run(sims, counts, 5)
    run(sims, counts, 64)
    sims_nan = sims.copy()
    sims_nan[100, 0] = sims_nan[7, 0] = np.nan                             # zero-norm rows: first, in event order
    run(sims_nan, counts, 5)
    run(sims[:2], np.array([1, 0]), 5)                                     # one hit in all: keep' = 1, the rest padded
    big = -np.sort(-rng.standard_normal((2000, k)).astype(np.float32), axis=1)      # 10 000 keys: three tournament windows
    big[1999, 0] = big[0, 0] = 7.5
    run(big, np.full(2000, k), 5)
    run(big, np.full(2000, k), 64)
    huge = -np.sort(-rng.standard_normal((14000, k)).astype(np.float32), axis=1)    # 70 000 keys: beyond the quick route (65 536)
    huge[13999, 0] = huge[1, 0] = 8.5
    run(huge, np.full(14000, k), 5)
    assert lib.hmm_rank_segment_hits(None, None, None, 1, 1, 1, None, None, None, None, None) == -1
    x = torch.zeros(8, device="cuda")
    assert lib.hmm_rank_segment_hits(x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 1, 65, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None) == -1
