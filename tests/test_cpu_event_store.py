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


# ---- the library's matrix parser (hmm_json_find_matrices / hmm_json_parse_matrix_f32) against json.load + np.array ----
def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(native, plain):
    nf, nt = native
    pf, pt = plain
    assert nf.keys() == pf.keys() and nt.keys() == pt.keys()
    for k in pf:
        assert nf[k].dtype == np.float32 and nf[k].shape == pf[k].shape and np.array_equal(_bits(nf[k]), _bits(pf[k])), k
    for k in pt:
        assert nt[k].dtype == pt[k].dtype and np.array_equal(nt[k], pt[k]), k


def test_native_matrix_parser_equals_json_load_on_the_golden_event_and_on_odd_values(tmp_path):
    """Same fp32 bits and same times as the reference's reading rule, on the golden file and on an event holding NaN / inf / -0.0 /
    subnormal / large / integer-valued entries and a matrix small enough to stay on json's path."""
    _same(es.parse_event_features(GOLD, native=True), es.parse_event_features(GOLD, native=False))
    rng = np.random.default_rng(11)
    v = rng.standard_normal((48, 1024)).astype(np.float32)
    v[0, :8] = [np.nan, np.inf, -np.inf, -0.0, 1e-42, -3.4e38, 16777217.0, 1e-7]
    case = dict(recipes.event_case(), features={"vision": v, "vision_times": np.arange(48) * 0.5,
                                                "audio": (rng.standard_normal((3, 1024)) * 20).astype(np.float32), "audio_times": [0.0, 10.0, 20.0],
                                                "tiny": [[0.25, 0.5], [1.0, 2.0]], "tiny_times": [0.0, 1.0]})
    p = es.save_event(case, tmp_path / "odd.json", write_sidecars=False)
    native = es.parse_event_features(p, native=True)
    _same(native, es.parse_event_features(p, native=False))
    assert np.array_equal(_bits(native[0]["vision"]), _bits(v))


def test_native_matrix_parser_old_format_strings_with_brackets_and_ragged_rows(tmp_path):
    rng = np.random.default_rng(12)
    v = rng.standard_normal((4, 1024)).astype(np.float32)
    ragged = [[1.0] * 600, [2.0] * 599]                                # not a matrix: stays a list, np.array() of it is the reference's business
    old = {"features": {"vision": {"features": v.tolist(), "times": [0.0, 1.0, 2.0, 3.0]}, "depth": v[:2].tolist()},
           "summary": 'a string with [[1.0, 2.0], [3.0, 4.0]] and a quote \\" inside ' + "[[" + "1.5, " * 2000 + "1.5]]",
           "frames": ["[[0.0]]"], "start_time": 0.0}
    p = tmp_path / "old.json"
    p.write_text(json.dumps(old, indent=2))
    _same(es.parse_event_features(p, native=True), es.parse_event_features(p, native=False))
    compact = tmp_path / "compact.json"                                # no whitespace at all, exponents, capital E
    compact.write_text(json.dumps(old, separators=(",", ":")).replace("e-", "E-"))
    _same(es.parse_event_features(compact, native=True), es.parse_event_features(compact, native=False))
    bad = {"features": {"vision": v.tolist()}, "feature_times": {"vision_times": [0.0, 1.0, 2.0, 3.0]}, "extra": [[0.5] * 40] * 40}
    p2 = tmp_path / "stray.json"                                       # a big matrix OUTSIDE features: the whole file takes json's path
    p2.write_text(json.dumps(bad))
    _same(es.parse_event_features(p2, native=True), es.parse_event_features(p2, native=False))


def test_native_number_conversion_is_float32_of_float64_of_the_text():
    """(float)(double)literal, bit for bit, on fp32 bit patterns, arbitrary doubles, and the decimals that sit on or next to a
    rounding boundary between two floats (where the library's short cut must hand over to the exact conversion)."""
    import ctypes as C
    from hippomm_amd import _lib as L
    lib = L.load()

    class Span(C.Structure):
        _fields_ = [("begin", C.c_size_t), ("end", C.c_size_t), ("rows", C.c_size_t), ("cols", C.c_size_t)]

    def native(texts, cols=64):
        rows = len(texts) // cols
        raw = ("[" + ",".join("[" + ",".join(texts[r * cols:(r + 1) * cols]) + "]" for r in range(rows)) + "]").encode()
        n, spans = C.c_int(0), (Span * 2)()
        L.check(lib.hmm_json_find_matrices(raw, len(raw), 1, C.cast(spans, C.c_void_p), 2, C.byref(n)), "find")
        assert n.value == 1 and (spans[0].rows, spans[0].cols, spans[0].begin, spans[0].end) == (rows, cols, 0, len(raw))
        a = np.empty((rows, cols), np.float32)
        L.check(lib.hmm_json_parse_matrix_f32(raw, 0, len(raw), rows, cols, a.ctypes.data_as(C.c_void_p)), "parse")
        return a, np.array(json.loads(raw), dtype=np.float64).astype(np.float32)

    rng = np.random.default_rng(13)
    lit = lambda xs, fmt=repr: [fmt(float(x)).replace("inf", "Infinity").replace("nan", "NaN") for x in xs]   # noqa: E731
    with np.errstate(all="ignore"):
        f32 = rng.integers(0, 2 ** 32, 64 * 600, dtype=np.uint64).astype(np.uint32).view(np.float32)
        f32 = f32[np.isfinite(f32)][:64 * 500]
        d64 = rng.integers(0, 2 ** 64, 64 * 600, dtype=np.uint64).view(np.float64)
        d64 = d64[np.isfinite(d64) & (np.abs(d64) < 1e300) & (np.abs(d64) > 1e-300)][:64 * 500]
        f = np.abs(rng.standard_normal(64 * 100).astype(np.float32)) * np.float32(10.0) ** rng.integers(-30, 30, 64 * 100).astype(np.float32)
        f = f[np.isfinite(f) & (f > 0)][:64 * 90]
        mid = (f.astype(np.float64) + np.nextafter(f, np.float32(np.inf)).astype(np.float64)) / 2
        special = [0.0, -0.0, 1.0, 5e-324, 1e-46, 1.4e-45, 7e-46, 3.4028234663852886e38, 3.4028235677973366e38, 3.5e38, 1e39, float("inf"),
                   float("-inf"), float("nan"), 1e22, 1e23, 123456789012345678.0, 0.1, 1 / 3, 1e-5, 16777217.0, 9007199254740993.0] * 64
        for tag, texts in (("fp32 bit patterns", lit(f32.astype(np.float64))), ("double bit patterns", lit(d64)),
                           ("midpoints", lit(mid)), ("midpoints + 1 ulp", lit(np.nextafter(mid, np.inf))),
                           ("midpoints - 1 ulp", lit(-np.nextafter(mid, -np.inf))), ("midpoints, 25 digits", lit(mid, lambda x: "%.24e" % x)),
                           ("specials", lit(special[:64 * 22]))):
            got, want = native(texts)
            assert np.array_equal(_bits(got), _bits(want)), tag
    got, want = native(["-0", "0", "7", "-12345678901"] * 16)
    assert np.array_equal(_bits(got), _bits(want))                     # "-0" is an int for json: +0.0


def test_native_parser_rejects_what_it_was_not_given():
    import ctypes as C
    from hippomm_amd import _lib as L
    lib = L.load()
    raw = b'{"a": [[1.0, 2.0], [3.0, 4.0]], "b": "x"}'
    out = np.empty((2, 2), np.float32)
    ptr = out.ctypes.data_as(C.c_void_p)
    assert lib.hmm_json_parse_matrix_f32(raw, 6, 30, 2, 2, ptr) == 0 and out.tolist() == [[1.0, 2.0], [3.0, 4.0]]
    assert lib.hmm_json_parse_matrix_f32(raw, 6, 30, 2, 3, ptr) != 0                  # wrong shape
    assert lib.hmm_json_parse_matrix_f32(raw, 0, 30, 2, 2, ptr) != 0                  # span starts at '{'
    assert lib.hmm_json_parse_matrix_f32(raw, 6, 6, 2, 2, ptr) != 0                   # empty span
    assert b"json_parse_matrix" in lib.hmm_last_error()
    huge = b"[[1e999, 1.0]]"                                                            # outside the double range: declined, not guessed
    assert lib.hmm_json_parse_matrix_f32(huge, 0, len(huge), 1, 2, ptr) != 0
    n = C.c_int(-1)
    assert lib.hmm_json_find_matrices(b'"[[1.0, 2.0]]" [[1.0], [2.0, 3.0]] [[]] [1.0]', 44, 1, None, 0, C.byref(n)) == 0 and n.value == 0


def test_build_event_store_reads_events_in_parallel_in_index_order(tmp_path, monkeypatch):
    """Threads must not reorder events: the loaded matrices line up with the index whatever finishes first."""
    rng = np.random.default_rng(14)
    index, want = {}, []
    for i in range(12):
        v = rng.standard_normal((3 + i, 1024)).astype(np.float32)
        ev = dict(recipes.event_case(), features={"vision": v, "vision_times": np.arange(3 + i) * 1.0})
        p = es.save_event(ev, tmp_path / "events" / "vid" / f"e{i}.json", write_sidecars=bool(i % 2))
        index[f"e{i}"] = {"file_path": str(p), "video_id": "vid"}
        want.append(v)
    (tmp_path / "event_index.json").write_text(json.dumps(index))
    got = []
    import hippomm_amd.vector_ops as vo
    monkeypatch.setattr(vo, "EventStore", lambda mats, device: got.extend(mats) or "store")
    store, ids = es.build_event_store(tmp_path, "vision", workers=4)
    assert store == "store" and ids == [f"e{i}" for i in range(12)]
    assert all(np.array_equal(_bits(a), _bits(b)) for a, b in zip(got, want))


def test_native_matrix_writer_is_json_dumps_indent_2_byte_for_byte():
    """hmm_json_write_matrix_f64 against json.dumps on fp32-origin values, arbitrary doubles, every power of ten, integers-valued
    floats and the notation switch points of float.__repr__ (1e16 / 1e15, 1e-4 / 1e-5), NaN / Infinity / -0.0."""
    rng = np.random.default_rng(15)

    def want(m):
        t = json.dumps({"a": {"b": m.tolist()}}, indent=2)           # closing bracket at indent 4, as features -> modality -> matrix
        return t[t.index("["):t.rindex("]") + 1]

    with np.errstate(all="ignore"):
        cases = [rng.integers(0, 2 ** 32, 64 * 800, dtype=np.uint64).astype(np.uint32).view(np.float32).astype(np.float64),
                 rng.integers(0, 2 ** 64, 64 * 800, dtype=np.uint64).view(np.float64), rng.standard_normal(64 * 400),
                 np.array([10.0 ** e for e in range(-323, 309)])[:64 * 9], rng.integers(-10 ** 17, 10 ** 17, 64 * 100).astype(np.float64),
                 np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 1e16, 9999999999999998.0, 1e15,
                           0.0001, 0.00001, 123456789012345680.0, 1234567.0, 0.1, 100.0, 1e22, 1e23, 1e-7, 1 / 3, 1e21, 9.999e-5, 1.5, -2.5e-10, 16777217.0] * 64)[:64 * 26]]
    for v in cases:
        m = v[: (len(v) // 64) * 64].reshape(-1, 64)
        assert es._matrix_text(m.tolist(), native=True) == want(m) == es._matrix_text(m.tolist(), native=False)
    assert es._matrix_text([[0.5]], native=True) == want(np.array([[0.5]]))


def test_event_files_are_read_and_written_without_the_library(tmp_path, monkeypatch):
    """Reading and saving an event is host work: with libhippomm_hip.so missing (or stale) the json route answers, same bytes and
    same values -- only the GPU paths refuse to run without the library."""
    from hippomm_amd import _lib, event_store as es
    rng = np.random.default_rng(8)
    feats = rng.standard_normal((40, 1024)).astype(np.float32)
    event = {"features": {"vision": feats, "vision_times": np.arange(40.0)}, "frames": [], "frame_times": [], "frame_captions": [],
             "audio_times": [], "audio_transcription": [], "holistic_audio_transcription": [], "summary": "s", "start_time": 0.0,
             "end_time": 39.0}
    with_lib = es.save_event(event, tmp_path / "a" / "ev.json", write_sidecars=False).read_bytes()

    def missing():
        raise _lib.HippoMMHipError("libhippomm_hip.so is missing")
    monkeypatch.setattr(_lib, "load", missing)
    without = es.save_event(event, tmp_path / "b" / "ev.json", write_sidecars=False).read_bytes()
    assert without == with_lib
    got, _ = es.parse_event_features(tmp_path / "b" / "ev.json")
    assert np.array_equal(got["vision"], feats)
