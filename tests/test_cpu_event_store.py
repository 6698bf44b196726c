"""CPU: memory_store event fast path against the JSON the reference's own ThetaEvent.to_dict
produced (tests/golden/event_golden.json, written by tests/golden/make_golden.py)."""
import json
import os
import time
from pathlib import Path

import numpy as np

import recipes
from hippomm_amd import event_store as es

GOLD = Path(__file__).resolve().parent / "golden" / "event_golden.json"


def test_writer_is_byte_identical_to_the_reference():
    case = recipes.event_case()
    assert es.event_json_text(case) == GOLD.read_text()
    d = es.event_to_dict(case)
    assert list(d) == ["features", "feature_times", "frames", "frame_times", "frame_captions", "audio_times",
                       "audio_transcription", "holistic_audio_transcription", "summary", "start_time", "end_time"]
    assert set(d["features"]) == {"vision", "audio"} and set(d["feature_times"]) == {"vision_times", "audio_times"}


def test_fast_writer_equals_json_dumps_indent_2():
    """The row-wise C-encoder path must give the bytes of json.dumps(to_dict(), indent=2) on anything an event can hold:
    the golden event, NaN / inf / -0 / denormal-range values, an empty matrix, a 1-D feature, integer-valued lists."""
    case = recipes.event_case()
    assert es.event_json_text(case, fast=True) == es.event_json_text(case, fast=False) == GOLD.read_text()
    rng = np.random.default_rng(5)
    v = rng.standard_normal((37, 1024)).astype(np.float32)
    v[3, 5], v[4, 5], v[7, 7], v[8, 7], v[9, 0] = np.nan, np.inf, 1e-30, -0.0, 123456792.0
    odd = dict(case, features={"vision": v, "vision_times": np.arange(37) * 0.5, "audio": np.zeros((0, 1024), np.float32),
                               "audio_times": [], "depth": rng.standard_normal(1024).astype(np.float32), "ints": [[1, 2], [3, 4]]})
    assert es.event_json_text(odd, fast=True) == json.dumps(es.event_to_dict(odd), indent=2)


class _DictEvent:
    """An event whose to_dict() returns an arbitrary dict: the fast writer must not trust what it finds in it."""

    def __init__(self, d):
        self._d = d

    def to_dict(self):
        return self._d


def test_fast_writer_checks_every_row_and_cannot_be_fooled_by_its_own_placeholder(monkeypatch):
    base = es.event_to_dict(recipes.event_case())
    rows = [[0.5, 1.5], [2.5, 3.5], [4.5, 5.5]]
    for bad_middle in ([2.5, "a, b"], [2.5, [1.0, 2.0]], [2.5, 3], [2.5, True]):        # a string with ", ", a nested list, an int, a bool
        d = dict(base, features={"vision": [rows[0], bad_middle, rows[2]]})
        assert es.event_json_text(_DictEvent(d), fast=True) == json.dumps(d, indent=2)
    # a string equal to the placeholder of matrix 0, earlier in the dict than the real hole: the writer must notice
    class _Fixed:
        hex = "0" * 32
    monkeypatch.setattr(es.uuid, "uuid4", lambda: _Fixed)
    token = "@@hmm_matrix_0_" + "0" * 32 + "@@"
    d = dict(base, features={"vision": rows})
    d = {"summary_first": token, **d}
    assert es.event_json_text(_DictEvent(d), fast=True) == json.dumps(d, indent=2)


def test_json_round_trip_is_exact_in_fp32():
    case = recipes.event_case()
    feats, times = es.parse_event_features(GOLD)
    for m in ("vision", "audio"):
        assert feats[m].dtype == np.float32
        np.testing.assert_array_equal(feats[m], case["features"][m])      # fp32 -> text -> fp32 is lossless
    np.testing.assert_array_equal(times["vision_times"], case["features"]["vision_times"])


def test_sidecar_is_used_when_fresh_and_rebuilt_when_stale(tmp_path):
    case = recipes.event_case()
    p = es.save_event(case, tmp_path / "events" / "vid" / "vid_0.json")
    assert p.read_text() == GOLD.read_text()
    npy = tmp_path / "events" / "vid" / "vid_0.vision.f32.npy"
    assert npy.exists() and (tmp_path / "events" / "vid" / "vid_0.sidecar.json").exists()
    # fresh: served from the sidecar (prove it by corrupting the sidecar content only)
    marker = case["features"]["vision"].copy(); marker[0, 0] = 42.0
    np.save(npy, marker)
    assert es.load_event_features(p)["vision"][0, 0] == 42.0
    # the JSON is rewritten by the reference (different mtime/size): sidecar is stale -> parsed again and repaired
    case2 = recipes.event_case(); case2["summary"] = "edited"
    time.sleep(0.01)
    p.write_text(es.event_json_text(case2))
    got = es.load_event_features(p)
    np.testing.assert_array_equal(got["vision"], case["features"]["vision"])
    np.testing.assert_array_equal(np.load(npy), case["features"]["vision"])
    # no sidecar at all (a store written by the unmodified reference)
    for f in p.parent.glob("vid_0.*"):
        if f != p:
            f.unlink()
    np.testing.assert_array_equal(es.load_event_features(p, write_sidecar=False)["audio"], case["features"]["audio"])
    assert not npy.exists()


def test_old_event_format_and_transposed_features(tmp_path):
    v = recipes.event_case()["features"]["vision"]
    old = {"features": {"vision": {"features": v.tolist(), "times": [0.0, 1.0, 2.5]}, "audio": v.T.tolist()}}
    p = tmp_path / "old.json"
    p.write_text(json.dumps(old))
    feats, times = es.parse_event_features(p)
    np.testing.assert_array_equal(feats["vision"], v)
    np.testing.assert_array_equal(feats["audio"], v)                       # (1024,3) is transposed back, :413-417
    assert times["vision"].tolist() == [0.0, 1.0, 2.5]


def test_sidecar_and_json_paths_agree_on_shape_and_survive_a_broken_sidecar(tmp_path):
    """save_event applies the same fix-ups as the JSON reader (a (1024,N) matrix is stored as (N,1024) either way), the
    loaded sidecar is checked against the manifest, and a truncated / foreign .npy falls back to the JSON."""
    case = recipes.event_case()
    v = case["features"]["vision"]
    case["features"]["vision"] = np.ascontiguousarray(np.tile(v, (2, 1)).T)          # (1024, 6): transposed on purpose
    p = es.save_event(case, tmp_path / "ev" / "t_0.json")
    fast = es.load_event_features(p)
    slow, _ = es.parse_event_features(p)
    assert fast["vision"].shape == slow["vision"].shape == (6, 1024)
    np.testing.assert_array_equal(fast["vision"], slow["vision"])
    npy = tmp_path / "ev" / "t_0.vision.f32.npy"
    good = npy.read_bytes()
    npy.write_bytes(good[: len(good) // 2])                                           # truncated: np.load raises ValueError
    np.testing.assert_array_equal(es.load_event_features(p, write_sidecar=False)["vision"], slow["vision"])
    np.save(npy, np.zeros((5, 1024), np.float64))                                     # foreign dtype / shape
    np.testing.assert_array_equal(es.load_event_features(p, write_sidecar=False)["vision"], slow["vision"])
    np.testing.assert_array_equal(es.load_event_features(p)["vision"], slow["vision"])   # repairs the cache
    assert np.load(npy).dtype == np.float32 and np.load(npy).shape == (6, 1024)


def test_index_iteration(tmp_path):
    case = recipes.event_case()
    base = tmp_path / "memory_store"
    index = {}
    for i in range(3):
        eid = f"vid_{i * 1000}"
        p = es.save_event(case, base / "events" / "vid" / f"{eid}.json", write_sidecars=(i != 1))
        index[eid] = {"video_id": "vid", "start_time": float(i), "end_time": float(i + 1), "file_path": str(p)}
    (base / "event_index.json").write_text(json.dumps(index, indent=2))
    got = list(es.iter_event_files(base))
    assert [g[0] for g in got] == list(index) and all(g[1].exists() for g in got)
