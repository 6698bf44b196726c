"""CPU: the host logic of hippomm_amd.vector_ops.enable_store_cache (the residency cache behind the reference's unchanged per-event
loop, hippocampal_memory.py:3143-3153) with a stand-in for the device store: fingerprint (every byte of event-sized arrays, a
sample above 64 MB), hit / miss, least-recently-used eviction by bytes, release when the host array dies, thread safety.  The GPU side of the same feature is tests/test_gpu_scan.py."""
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


def test_fingerprint_sees_identity_layout_and_every_value_of_event_sized_arrays():
    from hippomm_amd.vector_ops import _StoreCache
    cache = _StoreCache(1 << 30)
    rng = np.random.default_rng(0)
    a = rng.standard_normal((300, 1024)).astype(np.float32)
    fp = cache.fingerprint(a)
    assert fp == cache.fingerprint(a) and ("xxh3_64" in fp or "crc32" in fp)
    assert cache.fingerprint(a.copy()) != fp                                  # another buffer
    assert cache.fingerprint(a[:299]) != fp                                   # another shape
    assert cache.fingerprint(a.view(np.int32)) != fp                          # another dtype
    a[0, 0] += 1.0
    fp2 = cache.fingerprint(a)
    assert fp2 != fp
    a[157, 333] = np.nextafter(a[157, 333], np.float32(np.inf))               # one ulp, in an element no sample would visit
    assert cache.fingerprint(a) != fp2
    one_d = rng.standard_normal(1024).astype(np.float32)                      # a 1-D store is one row (vector_ops.py:173-174)
    assert cache.fingerprint(one_d) == cache.fingerprint(one_d)
    strided = a[::2]                                                          # non-contiguous view: hashed through a contiguous copy
    assert cache.fingerprint(strided) == cache.fingerprint(strided)


def test_arrays_above_the_limit_are_sampled_and_that_is_visible():
    """Above full_fingerprint_bytes only 64 rows x 16 columns are looked at: a sampled element is noticed, an unsampled one is
    not (the documented limitation of enable_store_cache for big stores)."""
    from hippomm_amd.vector_ops import _StoreCache
    cache = _StoreCache(1 << 30, full_fingerprint_bytes=1 << 20)
    a = np.random.default_rng(1).standard_normal((300, 1024)).astype(np.float32)   # 1.2 MB > 1 MB
    fp = cache.fingerprint(a)
    assert "sampled" in fp
    a[157, 333] += 1.0                                                        # not on the sampling grid
    assert cache.fingerprint(a) == fp
    a[0, 0] += 1.0                                                            # row 0 and column 0 are always sampled
    assert cache.fingerprint(a) != fp


def test_in_place_edit_of_an_unsampled_element_is_a_miss(monkeypatch):
    """Round-4 verdict item: with the old 64 x 16 sample this edit returned the stale resident copy silently."""
    from hippomm_amd import vector_ops as vo
    monkeypatch.setattr(vo, "FeatureStore", _FakeStore)
    _FakeStore.built = 0
    cache = vo._StoreCache(max_bytes=1 << 30)
    a = np.random.default_rng(2).standard_normal((600, 1024)).astype(np.float32)    # a 600-frame event: 2.4 MB
    s0 = cache.get(a)
    assert cache.get(a) is s0 and (cache.hits, cache.misses) == (1, 1)
    a[301, 517] = -a[301, 517]
    s1 = cache.get(a)
    assert s1 is not s0 and (cache.hits, cache.misses, _FakeStore.built) == (1, 2, 2)
    assert cache.get(a) is s1 and len(cache.entries) == 1


def test_cache_survives_concurrent_use(monkeypatch):
    import threading
    from hippomm_amd import vector_ops as vo
    monkeypatch.setattr(vo, "FeatureStore", _FakeStore)
    cache = vo._StoreCache(max_bytes=6 * 50 * 1024 * 4)
    arrays = [np.full((50, 1024), float(i), np.float32) for i in range(12)]
    errors = []

    def worker(seed):
        order = np.random.default_rng(seed).integers(0, len(arrays), 300)
        try:
            for i in order:
                cache.get(arrays[i])
        except Exception as ex:                                               # a KeyError from a racing eviction, before the lock
            errors.append(ex)
    threads = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors
    assert cache.bytes == sum(e[3] for e in cache.entries.values()) <= cache.max_bytes


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
