"""feature_search scan on MI355X -- host-side mirror of hippomm/utils/vector_ops.py.

``top_k_cosine_similarity(a, b, k)`` keeps the reference signature and return types
(vector_ops.py:151-188) and runs on the HIP kernels behind ``hmm_cosine_topk``.

The reference re-reads ``event.features[...]`` from host memory on every query
(hippocampal_memory.py:3153, :3304).  To keep the store resident in HBM between queries wrap
it once in a :class:`FeatureStore` (or pass a CUDA tensor); a numpy ``b`` is uploaded on every
call, which is correct but PCIe-bound.
"""
from __future__ import annotations

import threading
import weakref
import zlib
try:                                     # xxh3: ~17 GB/s on one core (2 MB event matrix: 0.1 ms); zlib's CRC-32 is the ~1 GB/s fallback
    from xxhash import xxh3_64_intdigest as _content_hash
    _CONTENT_HASH = "xxh3_64"
except ImportError:                      # pragma: no cover - the image ships xxhash
    _content_hash = zlib.crc32
    _CONTENT_HASH = "crc32"
from typing import List, Tuple, Union

import numpy as np
import torch

from . import _lib

FEATURE_DIM = 1024


class FeatureStore:
    """(N,1024) fp32 feature matrix resident in HBM (the ``memory_store`` vision/audio matrix
    of one event, or many events concatenated), plus the scan workspace."""

    def __init__(self, rows: Union[np.ndarray, torch.Tensor], device=None, shadow: bool = False):
        """``shadow=True`` also builds the bf16 shadow (+2048 B per row) and lets ``search`` / ``search_device`` -- and therefore
        ``top_k_cosine_similarity(q, store, k)`` -- go through it: the same results, bit for bit, for half the bytes streamed.
        The rows must stay unchanged while a shadow exists (see ``build_shadow``)."""
        dev = device or _lib.require_gpu()
        if isinstance(rows, np.ndarray):
            self.source_dtype = rows.dtype
            host = np.ascontiguousarray(rows.reshape(1, -1) if rows.ndim == 1 else rows, dtype=np.float32)
            t = torch.from_numpy(host).to(dev)
        else:
            self.source_dtype = np.dtype(str(rows.dtype).replace("torch.", "")) if rows.dtype in (
                torch.float32, torch.float64) else np.dtype(np.float32)
            t = rows.reshape(1, -1) if rows.dim() == 1 else rows
            t = t.to(device=dev, dtype=torch.float32).contiguous()
        if t.dim() != 2 or t.shape[1] != FEATURE_DIM:
            raise ValueError(f"store must be (N,{FEATURE_DIM}), got {tuple(t.shape)}")
        self.rows = t
        self._ws = None
        self._ws_key = None
        self.use_shadow = False
        if shadow and t.shape[0] > 0:
            self.build_shadow()
            self.use_shadow = True

    def __len__(self):
        return self.rows.shape[0]

    def _workspace(self, k: int):
        lib = _lib.load()
        need = lib.hmm_cosine_topk_workspace_bytes(len(self), k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.rows.device)
        return self._ws

    def search_device(self, query: torch.Tensor, k: int):
        """query: (1024,) fp32 CUDA tensor.  Returns CUDA tensors (idx int64[k'], sims fp32[k'])
        without synchronising (k' = min(k, N))."""
        if getattr(self, "use_shadow", False):
            return self.search_prefiltered_device(query, k)
        lib = _lib.load()
        n = len(self)
        if k < 1:
            raise ValueError("k must be >= 1")
        k_out = min(k, n)
        ws = self._workspace(k)
        idx = torch.empty(k_out, dtype=torch.int64, device=self.rows.device)
        sims = torch.empty(k_out, dtype=torch.float32, device=self.rows.device)
        n_out = torch.empty(1, dtype=torch.int32, device=self.rows.device)
        _lib.check(lib.hmm_cosine_topk(self.rows.data_ptr(), n, FEATURE_DIM, query.data_ptr(), k,
                                       idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                   "hmm_cosine_topk")
        return idx, sims

    def build_shadow(self, force: bool = False):
        """Build the bf16 shadow of the store (2048 B per row beside the 4096-B fp32 rows; hmm_shadow_store_build) that
        ``search_prefiltered_device`` streams.  Idempotent unless ``force``.

        The shadow is a SNAPSHOT: ``self.rows`` must not change while it exists (``FeatureStore`` / ``from_device_rows`` alias
        a caller's fp32 CUDA tensor without copying).  An in-place update made through torch bumps the tensor's version
        counter, which the searches check: the shadow is then rebuilt before it is used.  Writes that bypass torch (a raw
        kernel on ``rows.data_ptr()``) need ``invalidate_shadow()`` or ``build_shadow(force=True)``.  The build is enqueued
        on the current stream; searches on another stream must be ordered after it by the caller."""
        try:
            version = self.rows._version
        except (RuntimeError, AttributeError):       # an inference tensor tracks no version: a snapshot until invalidate_shadow()
            version = 0
        if force or getattr(self, "_shadow", None) is None or getattr(self, "_shadow_version", version) != version:
            lib = _lib.load()
            n = len(self)
            need = lib.hmm_shadow_store_bytes(n)
            if getattr(self, "_shadow", None) is None or self._shadow.numel() != need:
                self._shadow = torch.empty(need, dtype=torch.uint8, device=self.rows.device)
            _lib.check(lib.hmm_shadow_store_build(self.rows.data_ptr(), n, FEATURE_DIM, self._shadow.data_ptr(),
                                                  self._shadow.numel(), _lib.stream_ptr()), "hmm_shadow_store_build")
            self._shadow_version = version
        return self

    def invalidate_shadow(self):
        """Forget the bf16 shadow (after the fp32 rows were modified behind torch's back); the next prefiltered search
        rebuilds it."""
        self._shadow = None
        return self

    def search_prefiltered_device(self, query: torch.Tensor, k: int, stats: torch.Tensor = None):
        """``search_device`` through the bf16 shadow: candidates from one pass over 2048 B per row, re-scored exactly on the
        fp32 rows -- the same (idx, sims) as ``search_device``, bit for bit (hmm_cosine_topk_prefilter).  ``stats``: optional
        int32[2] CUDA tensor receiving (candidates re-scored, saturated lists)."""
        lib = _lib.load()
        self.build_shadow()
        n = len(self)
        if k < 1:
            raise ValueError("k must be >= 1")
        k_out = min(k, n)
        need = lib.hmm_cosine_topk_prefilter_workspace_bytes(n, k)
        if getattr(self, "_ws_pre", None) is None or self._ws_pre.numel() < need:
            self._ws_pre = torch.empty(need, dtype=torch.uint8, device=self.rows.device)
        idx = torch.empty(k_out, dtype=torch.int64, device=self.rows.device)
        sims = torch.empty(k_out, dtype=torch.float32, device=self.rows.device)
        n_out = torch.empty(1, dtype=torch.int32, device=self.rows.device)
        _lib.check(lib.hmm_cosine_topk_prefilter(self.rows.data_ptr(), self._shadow.data_ptr(), n, FEATURE_DIM, query.data_ptr(), k,
                                                 idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(),
                                                 stats.data_ptr() if stats is not None else None,
                                                 self._ws_pre.data_ptr(), self._ws_pre.numel(), _lib.stream_ptr()),
                   "hmm_cosine_topk_prefilter")
        return idx, sims

    def search_keys_device(self, query: torch.Tensor, k: int) -> torch.Tensor:
        """Local top-k as packed order keys (uint64 bit patterns in an int64 tensor, 0-padded to k)
        for the sharded scan (hippomm_amd.sharding.sharded_top_k)."""
        lib = _lib.load()
        ws = self._workspace(k)
        keys = torch.empty(k, dtype=torch.int64, device=self.rows.device)
        _lib.check(lib.hmm_cosine_topk_keys(self.rows.data_ptr(), len(self), FEATURE_DIM, query.data_ptr(), k,
                                            keys.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                   "hmm_cosine_topk_keys")
        return keys

    def search_segments_device(self, query: torch.Tensor, seg_offsets: torch.Tensor, k: int, prefilter: bool = False):
        """Per-event top-k in one pass.  seg_offsets: int64 CUDA tensor (E+1,), row offsets of the events inside
        this store.  Returns CUDA tensors idx (E,k) int64 rows within each event (-1 padded), sims (E,k) fp32,
        counts (E,) int32 = min(k, n_e).  ``prefilter``: stream the bf16 shadow (built on first use) and re-score each event's
        candidates on the fp32 rows -- the same outputs, bit for bit, for half the bytes (hmm_cosine_topk_segmented_prefilter)."""
        return self._search_segments_packed(query, seg_offsets, k, prefilter)[1:]

    def _search_segments_packed(self, query: torch.Tensor, seg_offsets: torch.Tensor, k: int, prefilter: bool = False):
        """search_segments_device, returning (packed, idx, sims, counts): the three outputs are views of `packed`."""
        lib = _lib.load()
        dev = self.rows.device
        E = seg_offsets.numel() - 1
        need = lib.hmm_cosine_topk_segmented_workspace_bytes(len(self), E, k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        # the three outputs are views of ONE buffer (idx | sims | counts), so that a caller who wants them on the host reads them back
        # with one copy (top_k_per_event)
        packed = torch.empty(E * k * 12 + E * 4, dtype=torch.uint8, device=dev)
        idx = packed[: E * k * 8].view(torch.int64).view(E, k)
        sims = packed[E * k * 8: E * k * 12].view(torch.float32).view(E, k)
        counts = packed[E * k * 12:].view(torch.int32)
        if prefilter and len(self) > 0:
            self.build_shadow()
            _lib.check(lib.hmm_cosine_topk_segmented_prefilter(self.rows.data_ptr(), self._shadow.data_ptr(), len(self), FEATURE_DIM,
                                                               query.data_ptr(), seg_offsets.data_ptr(), E, k, idx.data_ptr(),
                                                               sims.data_ptr(), counts.data_ptr(), self._ws.data_ptr(),
                                                               self._ws.numel(), _lib.stream_ptr()),
                       "hmm_cosine_topk_segmented_prefilter")
            return packed, idx, sims, counts
        _lib.check(lib.hmm_cosine_topk_segmented(self.rows.data_ptr(), len(self), FEATURE_DIM, query.data_ptr(),
                                                 seg_offsets.data_ptr(), E, k, idx.data_ptr(), sims.data_ptr(),
                                                 counts.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                 _lib.stream_ptr()), "hmm_cosine_topk_segmented")
        return packed, idx, sims, counts

    def search_multi_device(self, queries: torch.Tensor, k: int):
        """queries (Q,1024) fp32 on the store's device -> (idx (Q,k') int64, sims (Q,k') fp32) device tensors,
        k' = min(k, N).  One pass over the store per 16 queries (hmm_cosine_topk_multi)."""
        if queries.dim() != 2 or queries.shape[1] != FEATURE_DIM:
            raise ValueError(f"queries must be (Q,{FEATURE_DIM}), got {tuple(queries.shape)}")
        queries = queries.to(device=self.rows.device, dtype=torch.float32).contiguous()
        nq, n = queries.shape[0], len(self)
        if nq == 0:
            raise ValueError("no queries")
        lib = _lib.load()
        need = lib.hmm_cosine_topk_multi_workspace_bytes(n, nq, k)
        if self._ws is None or self._ws.numel() < need:      # kept between calls, like the single-query workspace
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.rows.device)
        ws = self._ws
        idx = torch.empty(nq, k, dtype=torch.int64, device=self.rows.device)
        sims = torch.empty(nq, k, dtype=torch.float32, device=self.rows.device)
        n_out = torch.empty(nq, dtype=torch.int32, device=self.rows.device)
        _lib.check(lib.hmm_cosine_topk_multi(self.rows.data_ptr(), n, FEATURE_DIM, queries.data_ptr(), nq, k,
                                                   idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), _lib.stream_ptr()), "hmm_cosine_topk_multi")
        kk = min(k, n)
        return idx[:, :kk], sims[:, :kk]

    def search_multi(self, queries, k: int) -> List[Tuple[np.ndarray, np.ndarray]]:
        """Per query the (indices int64, similarities) pair top_k_cosine_similarity would return."""
        q = queries if isinstance(queries, torch.Tensor) else torch.from_numpy(np.asarray(queries, dtype=np.float32))
        idx, sims = self.search_multi_device(q.reshape(-1, FEATURE_DIM), k)
        idx, sims = idx.cpu().numpy(), sims.cpu().numpy()
        return [(idx[i], sims[i]) for i in range(idx.shape[0])]

    def search(self, query, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _query_to_device(query, self.rows.device)
        idx, sims = self.search_device(q, k)
        return idx.cpu().numpy(), sims.cpu().numpy()


def merge_keys_device(keys: torch.Tensor, row_offsets: torch.Tensor, k: int):
    """keys: (n_shards, k) int64 CUDA (packed order keys); row_offsets: (n_shards,) int64 CUDA."""
    lib = _lib.load()
    n_shards = keys.shape[0]
    dev = keys.device
    idx = torch.empty(k, dtype=torch.int64, device=dev)
    sims = torch.empty(k, dtype=torch.float32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.hmm_topk_merge_keys(keys.contiguous().data_ptr(), n_shards, k, row_offsets.data_ptr(),
                                       idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(), _lib.stream_ptr()),
               "hmm_topk_merge_keys")
    n = int(n_out.item())
    return idx[:n], sims[:n]


# ---- optional residency for the UNCHANGED reference loop ----------------------------------------------------------------------
# The reference calls top_k_cosine_similarity(query, event.features['vision'], k=5) once per event and question with the event's
# host array (hippocampal_memory.py:3143-3153): as a drop-in that is a host-side fp32 conversion plus an upload per call (~1 ms
# per event).  With the cache enabled a numpy store seen before is served from its resident FeatureStore.  OFF by default.
# What it assumes: an array modified IN PLACE between two calls is noticed through a fingerprint of its contents -- a 64-bit
# hash of EVERY byte (xxh3, ~17 GB/s on one host core: 0.1 ms for a 500-frame event; CRC-32 when xxhash is absent) for arrays up
# to 64 MB (the reference's per-event matrices are 0.1-3 MB), and only a 64-row x 16-column sample (plus buffer address, shape,
# dtype, strides) above that, where a write to an unsampled element returns results from the stale HBM copy.
_STORE_CACHE = None
FULL_FINGERPRINT_BYTES = 64 << 20


class _StoreCache:
    def __init__(self, max_bytes: int, full_fingerprint_bytes: int = FULL_FINGERPRINT_BYTES):
        self.max_bytes, self.bytes, self.entries, self.hits, self.misses = int(max_bytes), 0, {}, 0, 0
        self.full_fingerprint_bytes = int(full_fingerprint_bytes)
        self._lock = threading.RLock()           # entries / bytes are also touched by weak-reference callbacks (any thread)

    def fingerprint(self, b: np.ndarray):
        head = (b.ctypes.data, b.shape, b.dtype.str, b.strides)
        if b.nbytes <= self.full_fingerprint_bytes:
            flat = b if b.flags.c_contiguous else np.ascontiguousarray(b)
            return head + (_CONTENT_HASH, _content_hash(memoryview(flat).cast("B")))
        rows = b.reshape(1, -1) if b.ndim == 1 else b
        pick = np.unique(np.linspace(0, rows.shape[0] - 1, num=min(rows.shape[0], 64)).astype(np.int64))
        return head + ("sampled", hash(np.ascontiguousarray(rows[pick][:, ::64]).tobytes()))

    def get(self, b: np.ndarray) -> "FeatureStore":
        key, fp = id(b), self.fingerprint(b)
        with self._lock:
            ent = self.entries.get(key)
            if ent is not None and ent[0]() is b and ent[2] == fp:
                self.hits += 1
                self.entries[key] = self.entries.pop(key)            # most recently used last
                return ent[1]
            if ent is not None:
                self._drop(key)
            self.misses += 1
        store = FeatureStore(b)
        size = store.rows.numel() * 4
        with self._lock:
            self._drop(key)                      # another thread may have inserted the same array while this one was uploading
            while self.entries and self.bytes + size > self.max_bytes:
                self._drop(next(iter(self.entries)))
            # the weak reference's callback releases the HBM copy when the host array dies
            self.entries[key] = (weakref.ref(b, lambda _r, k=key: self._drop(k)), store, fp, size)
            self.bytes += size
        return store

    def _drop(self, key):
        with self._lock:
            ent = self.entries.pop(key, None)
            if ent is not None:
                self.bytes -= ent[3]


def enable_store_cache(max_bytes: int = 8 << 30, full_fingerprint_bytes: int = FULL_FINGERPRINT_BYTES):
    """Keep the feature matrices that ``top_k_cosine_similarity`` receives as numpy arrays resident in HBM between calls (up to
    ``max_bytes`` of fp32 rows, least recently used first out): the reference's per-event loop then runs unchanged at the
    resident-store rate.

    Staleness: every call fingerprints the host array.  Arrays of at most ``full_fingerprint_bytes`` (default 64 MB; the
    reference's per-event matrices are 0.1-3 MB) are hashed WHOLE (xxh3 of every byte), so any in-place edit is a miss and
    the array is uploaded again.  Larger arrays are fingerprinted by a 64-row x 16-column sample plus address / shape / dtype /
    strides only: an in-place write to an unsampled element of such an array is NOT noticed and the answer comes from the
    stale resident copy -- wrap big stores in a ``FeatureStore`` yourself instead of relying on the cache."""
    global _STORE_CACHE
    _STORE_CACHE = _StoreCache(max_bytes, full_fingerprint_bytes)
    return _STORE_CACHE


def disable_store_cache():
    global _STORE_CACHE
    _STORE_CACHE = None


def _query_to_device(a, dev) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        q = a.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    else:
        q = torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.float32)).to(dev)
    if q.numel() != FEATURE_DIM:
        raise ValueError(f"query must have {FEATURE_DIM} elements, got {q.numel()}")
    return q


def top_k_cosine_similarity(
    a: Union[np.ndarray, torch.Tensor],
    b: Union[np.ndarray, torch.Tensor, FeatureStore],
    k: int,
) -> Tuple[np.ndarray, np.ndarray]:
    """Top-k cosine similarities between one vector and many (reference vector_ops.py:151-188).

    a: (1024,) query (numpy or torch); b: (N,1024) store (numpy, torch, or a resident
    FeatureStore; a 1-D b is one row, as at :173-174).  Returns (indices int64[k'],
    similarities[k']), k' = min(k, N) (k = 0: all N rows, k < 0: N - |k| rows, as the reference's slice gives), best first;
    similarities are float64 when either input was
    float64 (numpy's promotion at :182) and float32 otherwise.  Ties: higher row index first; a zero-norm
    row yields NaN and ranks first, as in the reference.  (The reference leaves the order INSIDE a tie group to numpy's
    unstable argsort; this total order is the library's own rule, include/hippomm_hip.h.)
    """
    a_is64 = (isinstance(a, np.ndarray) and a.dtype == np.float64) or (
        isinstance(a, torch.Tensor) and a.dtype == torch.float64)
    if not isinstance(b, FeatureStore) and getattr(b, "ndim", 2) == 2 and len(b) == 0:
        n, b_is64 = 0, str(getattr(b, "dtype", "")).endswith("float64")
    else:
        if isinstance(b, FeatureStore):
            store = b
        elif _STORE_CACHE is not None and isinstance(b, np.ndarray):
            store = _STORE_CACHE.get(b)
        else:
            store = FeatureStore(b)
        n, b_is64 = len(store), store.source_dtype == np.float64
    # the reference slices argsort(sims)[-k:][::-1] (:185): k = 0 keeps everything ([-0:]), k < 0 the best N - |k| rows
    k = int(k)
    k_eff = n if k == 0 else (max(n + k, 0) if k < 0 else min(k, n))
    out_dtype = np.float64 if (a_is64 or b_is64) else np.float32
    if k_eff == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=out_dtype)
    idx, sims = store.search(a, k_eff)
    return idx.astype(np.int64, copy=False), sims.astype(out_dtype, copy=False)


class EventStore(FeatureStore):
    """All events' feature matrices of one modality concatenated in HBM, with their row offsets: what
    ``QARecallSystem._find_relevant_*_segments`` iterates over (hippocampal_memory.py:3143, :3294)."""

    def __init__(self, event_features, device=None):
        mats = [np.ascontiguousarray(np.asarray(f).reshape(-1, FEATURE_DIM), dtype=np.float32) for f in event_features]
        self.lengths = [m.shape[0] for m in mats]
        rows = np.concatenate(mats, axis=0) if mats and sum(self.lengths) else np.zeros((0, FEATURE_DIM), np.float32)
        dev = device or _lib.require_gpu()
        if rows.shape[0] == 0:
            self.source_dtype = np.dtype(np.float32)
            self.rows = torch.zeros(0, FEATURE_DIM, dtype=torch.float32, device=dev)
            self._ws = None
        else:
            super().__init__(rows, dev)
        self.offsets = torch.tensor(np.concatenate([[0], np.cumsum(self.lengths)]), dtype=torch.int64, device=dev)

    @classmethod
    def from_device_rows(cls, rows: torch.Tensor, lengths) -> "EventStore":
        """An EventStore over a (N,1024) fp32 matrix that is already resident (events = consecutive row ranges of the
        given lengths); nothing is copied."""
        self = cls.__new__(cls)
        if rows.dim() != 2 or rows.shape[1] != FEATURE_DIM or rows.dtype != torch.float32 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous (N,1024) fp32 tensor")
        self.lengths = [int(n) for n in lengths]
        if sum(self.lengths) != rows.shape[0]:
            raise ValueError(f"event lengths sum to {sum(self.lengths)}, store has {rows.shape[0]} rows")
        self.source_dtype = np.dtype(np.float32)
        self.rows, self._ws, self._ws_key = rows, None, None
        self.offsets = torch.tensor(np.concatenate([[0], np.cumsum(self.lengths)]), dtype=torch.int64, device=rows.device)
        return self

    def top_hits(self, query, k: int = 5, keep: int = 5, prefilter: bool = False):
        """The caller's whole step in one go (hippocampal_memory.py:3143-3153 + :3275-3277): top-k per event, then every hit of
        every event ranked by similarity and the best `keep` returned as [(event index, row inside the event, similarity)].
        The ranking runs on the device (a stable descending sort of the (E, k) similarities in event order, i.e. what Python's
        stable ``sorted(..., reverse=True)`` does to the list the reference builds event by event); only `keep` hits are read
        back.  NaN similarities (zero-norm rows) rank first, as they do per event."""
        q = _query_to_device(query, self.rows.device)
        idx, sims, counts = self.search_segments_device(q, self.offsets, int(k), prefilter)
        E, keep = idx.shape[0], int(keep)
        if E == 0 or keep < 1:
            return []
        if keep > 64:                                           # beyond the ranking kernel's tournament width: torch's stable sort
            slot = torch.arange(idx.shape[1], device=idx.device).unsqueeze(0)
            valid = slot < counts.unsqueeze(1)
            key = torch.where(valid, torch.nan_to_num(sims, nan=float("inf")), torch.full_like(sims, float("-inf"))).reshape(-1)
            order = torch.sort(key, descending=True, stable=True).indices[:keep]
            order = order[valid.reshape(-1)[order]]
            flat_idx, flat_sims = idx.reshape(-1)[order], sims.reshape(-1)[order]
            hits = torch.stack([(order // idx.shape[1]).double(), flat_idx.double(), flat_sims.double()]).cpu().numpy()
            return [(int(e), int(r), float(np.float32(v))) for e, r, v in zip(hits[0], hits[1], hits[2])]
        # one launch ranks the (E, k) hits (hmm_rank_segment_hits), one copy brings the `keep` best back: event | row | sim | count
        lib = _lib.load()
        packed = torch.empty(keep * 20 + 4, dtype=torch.uint8, device=idx.device)
        ev = packed[: keep * 8].view(torch.int64)
        row = packed[keep * 8: keep * 16].view(torch.int64)
        val = packed[keep * 16: keep * 20].view(torch.float32)
        n_out = packed[keep * 20:].view(torch.int32)
        _lib.check(lib.hmm_rank_segment_hits(idx.data_ptr(), sims.data_ptr(), counts.data_ptr(), E, idx.shape[1], keep, ev.data_ptr(),
                                             row.data_ptr(), val.data_ptr(), n_out.data_ptr(), _lib.stream_ptr()), "hmm_rank_segment_hits")
        raw = packed.cpu().numpy()
        n = int(raw[keep * 20:].view(np.int32)[0])
        ev_h, row_h, val_h = raw[: keep * 8].view(np.int64), raw[keep * 8: keep * 16].view(np.int64), raw[keep * 16: keep * 20].view(np.float32)
        return [(int(ev_h[t]), int(row_h[t]), float(val_h[t])) for t in range(n)]

    def top_k_per_event(self, query, k: int = 5, prefilter: bool = False):
        """[(indices int64[k_e], sims float32[k_e]) for every event], each exactly what
        ``top_k_cosine_similarity(query, event_features, k)`` returns for that event."""
        q = _query_to_device(query, self.rows.device)
        packed, idx, _, _ = self._search_segments_packed(q, self.offsets, int(k), prefilter)
        E, k = idx.shape
        if E == 0:
            return []
        # one copy of the packed (idx | sims | counts) buffer into pinned memory instead of three synchronising .cpu() calls
        host = self._readback_buffer(packed.numel())
        host.copy_(packed, non_blocking=True)
        torch.cuda.current_stream(packed.device).synchronize()
        raw = host.numpy()
        idx_h = raw[: E * k * 8].view(np.int64).reshape(E, k).copy()
        sims_h = raw[E * k * 8: E * k * 12].view(np.float32).reshape(E, k).copy()
        counts_h = raw[E * k * 12:].view(np.int32)
        if int(counts_h.min()) == k:                             # every event has k rows: plain row views, no slicing
            return list(zip(idx_h, sims_h))
        return [(idx_h[e, :counts_h[e]], sims_h[e, :counts_h[e]]) for e in range(E)]

    def _readback_buffer(self, nbytes: int) -> torch.Tensor:
        buf = getattr(self, "_pinned", None)
        if buf is None or buf.numel() < nbytes:
            buf = self._pinned = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True)
        return buf[:nbytes]
