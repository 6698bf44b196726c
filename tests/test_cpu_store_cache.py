"""CPU: the host logic of hippomm_amd.vector_ops.enable_store_cache (the residency cache behind the reference's unchanged per-event
loop, hippocampal_memory.py:3143-3153) with a stand-in for the device store: fingerprint, hit / miss, least-recently-used eviction by
bytes, release when the host array dies.  The GPU side of the same feature is tests/test_gpu_scan.py."""
import gc

import numpy as np


class _FakeRows:
    def __init__(self, n):
        self._n = n

    def numel(self):
        return self._n


class _FakeStore:
    built = 0

    def __init__(self, b):
        type(self).built += 1
        self.rows = _FakeRows(int(np.asarray(b).size))


def test_fingerprint_sees_identity_layout_and_sampled_values():
    from hippomm_amd.vector_ops import _StoreCache
    rng = np.random.default_rng(0)
    a = rng.standard_normal((300, 1024)).astype(np.float32)
    fp = _StoreCache.fingerprint(a)
    assert fp == _StoreCache.fingerprint(a)
    assert _StoreCache.fingerprint(a.copy()) != fp                            # another buffer
    assert _StoreCache.fingerprint(a[:299]) != fp                             # another shape
    assert _StoreCache.fingerprint(a.view(np.int32)) != fp                    # another dtype
    a[0, 0] += 1.0                                                            # row 0 and column 0 are always sampled
    assert _StoreCache.fingerprint(a) != fp
    one_d = rng.standard_normal(1024).astype(np.float32)                      # a 1-D store is one row (vector_ops.py:173-174)
    assert _StoreCache.fingerprint(one_d) == _StoreCache.fingerprint(one_d)


def test_hits_evictions_and_release(monkeypatch):
    from hippomm_amd import vector_ops as vo
    monkeypatch.setattr(vo, "FeatureStore", _FakeStore)
    _FakeStore.built = 0
    cache = vo._StoreCache(max_bytes=3 * 100 * 1024 * 4)
    arrays = [np.full((100, 1024), float(i), np.float32) for i in range(4)]
    s0 = cache.get(arrays[0])
    assert cache.get(arrays[0]) is s0 and (cache.hits, cache.misses, _FakeStore.built) == (1, 1, 1)
    cache.get(arrays[1]); cache.get(arrays[2])
    assert cache.bytes == 3 * 100 * 1024 * 4 and len(cache.entries) == 3
    cache.get(arrays[0])                                                      # touch 0: 1 is now the least recently used
    cache.get(arrays[3])                                                      # does not fit beside three: evicts 1
    assert id(arrays[1]) not in cache.entries and id(arrays[0]) in cache.entries and len(cache.entries) == 3
    arrays[2][0, 0] = 99.0                                                    # modified in place where the fingerprint looks
    before = _FakeStore.built
    cache.get(arrays[2])
    assert _FakeStore.built == before + 1 and cache.bytes == 3 * 100 * 1024 * 4
    gone = id(arrays[3])
    del arrays[3]
    gc.collect()
    assert gone not in cache.entries and cache.bytes == 2 * 100 * 1024 * 4    # released with the host array


def test_cache_is_off_unless_enabled():
    from hippomm_amd import vector_ops as vo
    assert vo._STORE_CACHE is None
    c = vo.enable_store_cache(1 << 20)
    assert vo._STORE_CACHE is c
    vo.disable_store_cache()
    assert vo._STORE_CACHE is None
