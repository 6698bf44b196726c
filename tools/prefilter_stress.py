"""Soak of the bf16-prefilter scan: many queries against a random 1M-row store, a clustered store (scenes of near-identical rows), a
store with exact duplicates and NaN rows, and per-event scans -- every answer must be bit-identical to the exact scan's, with a large
GEMM on a second stream perturbing the timing.  usage: prefilter_stress.py [queries per store]   (product library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippomm_amd import _lib as L
from hippomm_amd.vector_ops import EventStore, FeatureStore

lib = L.load()
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 200
side = torch.cuda.Stream()
big_a = torch.randn(8192, 1280, device="cuda").to(torch.bfloat16)
big_w = torch.randn(5120, 1280, device="cuda").to(torch.bfloat16)
big_b = torch.zeros(5120, device="cuda")
big_c = torch.empty(8192, 5120, dtype=torch.bfloat16, device="cuda")
g = torch.Generator(device="cuda").manual_seed(7)
bad = 0
fallbacks = 0


def stores():
    n = 1_000_000
    rows = torch.randn(n, 1024, device="cuda", generator=g)
    yield "random 1M", rows
    scenes = torch.randn(5000, 1024, device="cuda", generator=g)
    rows = scenes.repeat_interleave(60, dim=0) + 2e-3 * torch.randn(300_000, 1024, device="cuda", generator=g)
    yield "scenes of 60 near-identical rows", rows
    base = torch.randn(64, 1024, device="cuda", generator=g)
    rows = base[torch.randint(0, 64, (200_000,), device="cuda", generator=g)].clone()
    rows[::40_000] = 0.0
    yield "exact duplicates + NaN rows", rows


for name, rows in stores():
    fs = FeatureStore(rows).build_shadow()
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    n_ev = 500
    es = EventStore.from_device_rows(fs.rows, [fs.rows.shape[0] // n_ev] * n_ev)
    es._shadow = fs._shadow
    for i in range(Q):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                L.check(lib.hmm_op_gemm_bf16_tile(big_a.data_ptr(), big_w.data_ptr(), big_b.data_ptr(), big_c.data_ptr(), 8192, 5120, 1280, 1, 3,
                                                  side.cuda_stream), "load")
        k = (1, 5, 32, 64)[i % 4]
        if i % 3 == 0:
            q = fs.rows[int(torch.randint(0, fs.rows.shape[0], (1,)))] + 0.2 * torch.randn(1024, device="cuda", generator=g)
        else:
            q = torch.randn(1024, device="cuda", generator=g)
        i0, s0 = fs.search_device(q, k)
        i1, s1 = fs.search_prefiltered_device(q, k, stats)
        ok = torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
        st = stats.cpu().tolist()
        fallbacks += int(st[1] > 0 or st[0] > 1024)
        if i % 10 == 0:
            a, b = es.search_segments_device(q, es.offsets, 5), es.search_segments_device(q, es.offsets, 5, prefilter=True)
            ok = ok and torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)) and torch.equal(a[2], b[2])
        if not ok:
            bad += 1
            print(f"{name}: query {i} k={k} DIFFERS (stats {st})", flush=True)
    torch.cuda.synchronize()
    print(f"{name}: {Q} queries done, exact-scan fallbacks so far {fallbacks}", flush=True)
    del fs, es
    torch.cuda.empty_cache()
print("prefilter stress:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
