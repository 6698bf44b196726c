"""memory_store event files: fast path beside the reference's JSON.

The reference stores every consolidated event as ``events/<video_id>/<event_id>.json`` written by
``json.dump(event.to_dict(), f, indent=2)`` (hippomm/core/hippocampal_memory.py:110-133, :331-335) and reads
it back with ``json.load`` + ``np.array(list)`` (:369-395): one Python float per line, ~20 bytes of text per
stored fp32, float64 arrays after loading.  That file format is the contract with the unmodified
``load_theta_event`` and is kept byte for byte.  What this module adds:

* ``save_event``     writes the same JSON **plus** an fp32 ``.npy`` sidecar per feature matrix and a manifest
                     holding the JSON's size and mtime;
* ``load_event_features``  returns the fp32 matrices from the sidecars when they are fresh (same size+mtime
                     as recorded), else parses the JSON once and (re)writes the sidecars.  Values are
                     identical either way: the JSON holds fp32 values printed as doubles, and
                     ``float32(float64(text))`` round-trips them exactly;
* ``build_event_store``  reads ``event_index.json`` (:338-346) and returns an ``EventStore`` with one modality's
                     matrices of all events resident in HBM, ready for ``top_k_per_event``.

Nothing here touches the GPU except ``build_event_store``.
"""
from __future__ import annotations

import json
import os
import uuid
from pathlib import Path
from typing import Any, Dict, Iterable, Mapping, Optional, Tuple

import numpy as np

SIDECAR_VERSION = 1
_EVENT_KEYS = ("frames", "frame_times", "frame_captions", "audio_times", "audio_transcription",
               "holistic_audio_transcription", "summary", "start_time", "end_time")


def event_to_dict(event: Any) -> Dict[str, Any]:
    """The dict ``ThetaEvent.to_dict`` builds (:110-133), from a ThetaEvent-like object or a mapping with the
    same fields.  Keys ending in ``_times`` inside ``features`` move to ``feature_times``; arrays become lists."""
    get = (lambda k, d=None: event.get(k, d)) if isinstance(event, Mapping) else (lambda k, d=None: getattr(event, k, d))
    features_dict, times_dict = {}, {}
    for modality, features in get("features").items():
        (times_dict if modality.endswith("_times") else features_dict)[modality] = np.asarray(features).tolist()
    return {
        "features": features_dict, "feature_times": times_dict,
        "frames": get("frames"), "frame_times": get("frame_times"), "frame_captions": get("frame_captions"),
        "audio_times": get("audio_times"), "audio_transcription": get("audio_transcription"),
        "holistic_audio_transcription": get("holistic_audio_transcription"), "summary": get("summary"),
        "start_time": get("start_time"), "end_time": get("end_time"),
    }


def event_json_text(event: Any, fast: bool = True) -> str:
    """Exactly the text ``save_theta_event`` writes (:334-335): ``json.dumps(event.to_dict(), indent=2)``.

    ``indent`` forces Python's pure-Python encoder (~1 us per float; 0.6 s for a 600-frame event).  With ``fast`` the
    2-D feature matrices -- 99.9 % of the text -- are written by the library (``_matrix_text``: the same bytes -- same
    ``float.__repr__`` digits and notation, same separators and indentation); everything else, and any matrix that is not a non-empty list of equally long rows of Python floats (EVERY row is
    checked), goes through ``json.dumps(indent=2)`` itself.  The hole a matrix leaves in the outer text is a fresh random
    token; if the token turns up anywhere else in the text the whole event goes through ``json.dumps(indent=2)``."""
    d = event.to_dict() if hasattr(event, "to_dict") else event_to_dict(event)
    if not fast or not isinstance(d.get("features"), dict):
        return json.dumps(d, indent=2)
    feats, holes, salt = dict(d["features"]), {}, uuid.uuid4().hex
    for i, (modality, rows) in enumerate(list(feats.items())):
        if (isinstance(rows, list) and rows and isinstance(rows[0], list) and rows[0]
                and all(type(r) is list and len(r) == len(rows[0]) and set(map(type, r)) == {float} for r in rows)):   # every value a float
            token = f"@@hmm_matrix_{i}_{salt}@@"
            holes[f'"{token}"'] = rows
            feats[modality] = token
    text = json.dumps(dict(d, features=feats), indent=2)
    if any(text.count(quoted) != 1 for quoted in holes):         # the token also occurs in the event's own strings
        return json.dumps(d, indent=2)
    for quoted, rows in holes.items():
        # features -> modality -> row -> value: rows sit at indent 6, values at indent 8, the closing bracket at indent 4
        text = text.replace(quoted, _matrix_text(rows), 1)
    return text


def _host_library():
    """The library for its HOST-side JSON routines, or None when it is not built / too old: reading and saving an event file is
    plain host work and keeps working on json alone (the GPU paths have no such fallback and raise)."""
    from . import _lib as L
    try:
        return L.load()
    except (L.HippoMMHipError, AttributeError, OSError):
        return None


def _matrix_text(rows, native: bool = True) -> str:
    """``json.dumps(rows, indent=2)`` of a non-empty list of equally long lists of floats whose closing bracket sits at indent 4.
    ``native``: written by the library (``hmm_json_write_matrix_f64``: float.__repr__ digits and notation, byte for byte --
    tests/test_cpu_event_store.py); otherwise row by row with the C encoder of ``json`` and re-indented."""
    lib = _host_library() if native else None
    if lib is not None:
        import ctypes as C
        from . import _lib as L
        m = np.array(rows, dtype=np.float64)
        cap = lib.hmm_json_matrix_text_bound(m.shape[0], m.shape[1], 4)
        buf = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        L.check(lib.hmm_json_write_matrix_f64(m.ctypes.data_as(C.c_void_p), m.shape[0], m.shape[1], 4, C.cast(buf, C.c_void_p), cap, C.byref(n)),
                "hmm_json_write_matrix_f64")
        return C.string_at(buf, n.value).decode("ascii")          # only the written bytes (buf.raw would copy the whole bound first)
    body = ",\n".join("      [\n        " + json.dumps(r)[1:-1].replace(", ", ",\n        ") + "\n      ]" for r in rows)
    return "[\n" + body + "\n    ]"


def _sidecar_paths(json_path: Path, modality: str) -> Tuple[Path, Path]:
    stem = json_path.with_suffix("")
    return Path(f"{stem}.{modality}.f32.npy"), Path(f"{stem}.sidecar.json")


def _as_feature_matrix(a) -> np.ndarray:
    """One rule for both paths (JSON parse and sidecar): fp32, C-contiguous, and the reference's shape fix-up when a
    matrix arrives as (1024, N) (:413-417) -- so that an event loads with the same shape whichever path serves it."""
    a = np.asarray(a)
    if a.ndim > 1 and a.shape[1] != 1024 and a.shape[0] == 1024:
        a = a.T
    return np.ascontiguousarray(a, dtype=np.float32)


def _write_sidecars(json_path: Path, features: Mapping[str, np.ndarray]) -> None:
    st = json_path.stat()
    shapes = {}
    for modality, arr in features.items():
        a = _as_feature_matrix(arr)
        npy, _ = _sidecar_paths(json_path, modality)
        tmp = npy.with_suffix(".tmp.npy")
        np.save(tmp, a)
        os.replace(tmp, npy)
        shapes[modality] = list(a.shape)
    _, manifest = _sidecar_paths(json_path, "x")
    tmp = manifest.with_suffix(".tmp")
    tmp.write_text(json.dumps({"version": SIDECAR_VERSION, "json_size": st.st_size, "json_mtime_ns": st.st_mtime_ns,
                               "modalities": shapes}))
    os.replace(tmp, manifest)


def save_event(event: Any, json_path, write_sidecars: bool = True) -> Path:
    """Write the reference's JSON (byte-identical) and, optionally, the fp32 sidecars next to it."""
    json_path = Path(json_path)
    json_path.parent.mkdir(parents=True, exist_ok=True)
    json_path.write_text(event_json_text(event))
    if write_sidecars:
        feats = event.features if hasattr(event, "features") else event["features"]
        _write_sidecars(json_path, {m: f for m, f in feats.items() if not m.endswith("_times")})
    return json_path


def _fresh_manifest(json_path: Path) -> Optional[dict]:
    _, manifest = _sidecar_paths(json_path, "x")
    try:
        m = json.loads(manifest.read_text())
        st = json_path.stat()
        if m.get("version") == SIDECAR_VERSION and m["json_size"] == st.st_size and m["json_mtime_ns"] == st.st_mtime_ns:
            return m
    except (OSError, ValueError, KeyError):
        pass
    return None


NATIVE_MIN_VALUES = 1024        # 2-D arrays of at least this many numbers are parsed by the library, the rest by json


class _NativeMatrix:
    """Placeholder json.loads leaves where the library parsed a matrix."""
    __slots__ = ("array",)

    def __init__(self, array):
        self.array = array


def _load_json_native(raw: bytes):
    """``json.loads(raw)`` with every large 2-D array of numbers parsed by the library (``hmm_json_find_matrices`` /
    ``hmm_json_parse_matrix_f32``: fp32, the value ``np.array(list).astype(float32)`` has) and left in the result as a
    ``_NativeMatrix``.  None when the library declines (a literal outside the double range) or a matrix sits where this module
    does not expect one -- the caller then uses plain ``json.loads``."""
    import ctypes as C
    from . import _lib as L
    lib = _host_library()
    if lib is None:
        return None

    class Span(C.Structure):
        _fields_ = [("begin", C.c_size_t), ("end", C.c_size_t), ("rows", C.c_size_t), ("cols", C.c_size_t)]

    n = C.c_int(0)
    cap = 16
    while True:
        spans = (Span * cap)()
        L.check(lib.hmm_json_find_matrices(raw, len(raw), NATIVE_MIN_VALUES, C.cast(spans, C.c_void_p), cap, C.byref(n)), "hmm_json_find_matrices")
        if n.value <= cap:
            break
        cap = n.value
    if n.value == 0:
        return json.loads(raw)
    salt = uuid.uuid4().hex
    pieces, mats, pos = [], {}, 0
    for i in range(n.value):
        sp = spans[i]
        a = np.empty((sp.rows, sp.cols), np.float32)
        if lib.hmm_json_parse_matrix_f32(raw, sp.begin, sp.end, sp.rows, sp.cols, a.ctypes.data_as(C.c_void_p)) != 0:
            return None
        token = f"@@hmm_matrix_{i}_{salt}@@"
        mats[token] = a
        pieces += [raw[pos:sp.begin], b'"' + token.encode() + b'"']
        pos = sp.end
    pieces.append(raw[pos:])
    reduced = b"".join(pieces)
    if any(reduced.count(t.encode()) != 1 for t in mats):          # the token also occurs in the event's own strings
        return None

    def put_back(v):
        if isinstance(v, str) and v in mats:
            return _NativeMatrix(mats[v])
        if isinstance(v, dict):
            return {k: put_back(x) for k, x in v.items()}
        if isinstance(v, list):
            return [put_back(x) for x in v]
        return v
    return put_back(json.loads(reduced))


def parse_event_features(json_path, native: bool = True) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """An event file without sidecars, by the reference's own reading rules (:369-408): new format with ``feature_times``, and
    the old format where a modality maps to ``{'features': ..., 'times': ...}``.  Features come back as fp32.  With ``native`` the
    feature matrices -- 99.9 % of the text -- are converted by the library instead of ``json.load`` + ``np.array(list)`` (~10x
    faster, same values: tests/test_cpu_event_store.py); ``native=False`` is the reference's path as it stands."""
    raw = Path(json_path).read_bytes()
    data = _load_json_native(raw) if native else None
    if data is not None:
        # matrices are expected as feature values only; anywhere else (times, metadata) keeps the reference's float64 lists
        feats_in = data.get("features") if isinstance(data, dict) else None
        allowed = set()
        if isinstance(feats_in, dict):
            for d in feats_in.values():
                if isinstance(d, _NativeMatrix):
                    allowed.add(id(d))
                elif isinstance(d, dict) and isinstance(d.get("features"), _NativeMatrix):
                    allowed.add(id(d["features"]))

        def stray(v):
            if isinstance(v, _NativeMatrix):
                return id(v) not in allowed
            if isinstance(v, dict):
                return any(stray(x) for x in v.values())
            if isinstance(v, list):
                return any(stray(x) for x in v)
            return False
        if stray(data):
            data = None
    if data is None:
        data = json.loads(raw)
    unwrap = lambda v: v.array if isinstance(v, _NativeMatrix) else np.array(v)     # noqa: E731
    feats, times = {}, {}
    if "feature_times" in data:
        for modality, t in data["feature_times"].items():
            times[modality] = np.array(t)
        for modality, f in data["features"].items():
            feats[modality] = unwrap(f)
    else:
        for modality, d in data["features"].items():
            if isinstance(d, dict):
                if "features" in d:
                    feats[modality] = unwrap(d["features"])
                if "times" in d:
                    times[modality] = np.array(d["times"])
            else:
                feats[modality] = unwrap(d)
    return {modality: _as_feature_matrix(a) for modality, a in feats.items()}, times


def load_event_features(json_path, use_sidecar: bool = True, write_sidecar: bool = True,
                        mmap: bool = False) -> Dict[str, np.ndarray]:
    """fp32 feature matrices of one event; sidecar when fresh, JSON otherwise."""
    json_path = Path(json_path)
    if use_sidecar:
        m = _fresh_manifest(json_path)
        if m is not None:
            try:
                out = {}
                for mod, shape in m["modalities"].items():
                    a = np.load(_sidecar_paths(json_path, mod)[0], mmap_mode="r" if mmap else None)
                    if a.dtype != np.float32 or list(a.shape) != list(shape):
                        raise ValueError(f"sidecar of {mod!r} is {a.dtype}{a.shape}, manifest says float32{tuple(shape)}")
                    out[mod] = a
                return out
            except (OSError, ValueError):            # missing, truncated or foreign .npy: fall back to the JSON
                pass
    feats, _ = parse_event_features(json_path)
    if write_sidecar:
        try:
            _write_sidecars(json_path, feats)
        except OSError:
            pass                                    # read-only store: keep working from the JSON
    return feats


def iter_event_files(memory_store_dir) -> Iterable[Tuple[str, Path]]:
    """(event_id, json path) in index order, from ``<base>/event_index.json`` (:338-346)."""
    base = Path(memory_store_dir)
    index = json.loads((base / "event_index.json").read_text())
    for event_id, info in index.items():
        p = Path(info["file_path"])
        if not p.is_absolute() and not p.exists():
            p = base / "events" / info["video_id"] / f"{event_id}.json"
        yield event_id, p


def build_event_store(memory_store_dir, modality: str = "vision", device=None, workers: Optional[int] = None):
    """All events' ``modality`` matrices of a memory_store, resident in HBM.  Returns (EventStore, event_ids);
    events that lack the modality or whose width is not 1024 get an empty segment (the reference skips them
    with a warning, :3135-3137).  Event files are read by ``workers`` threads (default: min(16, cores)): the sidecar reads and
    the library's matrix parser run outside the GIL, so a cold store loads in parallel; the order of the events is the index's."""
    from concurrent.futures import ThreadPoolExecutor
    from .vector_ops import EventStore
    files = list(iter_event_files(memory_store_dir))
    if workers is None:
        workers = min(16, os.cpu_count() or 1)
    if workers > 1 and len(files) > 1:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            loaded = list(pool.map(lambda f: load_event_features(f[1]), files))
    else:
        loaded = [load_event_features(path) for _, path in files]
    ids, mats = [], []
    for (event_id, path), feats in zip(files, loaded):
        a = feats.get(modality)
        if a is not None and a.ndim == 1 and a.shape[0] == 1024:
            a = a.reshape(1, 1024)                   # top_k_cosine_similarity treats a 1-D feature as one row (vector_ops.py:173-174)
        if a is None or a.ndim != 2 or a.shape[1] != 1024:
            a = np.zeros((0, 1024), np.float32)
        ids.append(event_id)
        mats.append(a)
    return EventStore(mats, device), ids
