"""Same session, same store: the streaming kernels of round 5's library (tools/libhippomm_r5.so, built from commit 09ccf69) against
this round's, interleaved -- did the refactored scan body / the new tournament change the streaming rate?  (The pool's boxes differ
by +- 5 % on HBM-bound kernels, so numbers from different sessions cannot answer that.)
    python tools/scan_ab_r5_probe.py [out.json] [libhippomm_<variant>.so ...]      (extra builds under tools/, timed beside the two)

tools/libhippomm_r5.so is not tracked: `mkdir /tmp/r5 && git archive 09ccf69 | tar -x -C /tmp/r5 && (cd /tmp/r5 && python -m hippomm_amd.build)
&& cp /tmp/r5/hippomm_amd/libhippomm_hip.so tools/libhippomm_r5.so`."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from hippomm_amd import _lib as L
from hippomm_amd.vector_ops import FeatureStore

new = L.load()
old = C.CDLL(os.path.join(ROOT, "tools", "libhippomm_r5.so"))
extra = {os.path.basename(a)[len("libhippomm_"):-3]: C.CDLL(os.path.join(ROOT, "tools", a)) for a in sys.argv[1:] if a.endswith(".so")}
for lib in (old, *extra.values()):
    lib.hmm_op_scan_topk_only.restype = C.c_int
    lib.hmm_op_scan_topk_only.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.hmm_cosine_topk_prefilter.restype = C.c_int
    lib.hmm_cosine_topk_prefilter.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_void_p]
    lib.hmm_cosine_topk.restype = C.c_int
    lib.hmm_cosine_topk.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.hmm_cosine_topk_prefilter_workspace_bytes.restype = C.c_size_t
    lib.hmm_cosine_topk_prefilter_workspace_bytes.argtypes = [C.c_int64, C.c_int]
    lib.hmm_cosine_topk_segmented_workspace_bytes.restype = C.c_size_t
    lib.hmm_cosine_topk_segmented_workspace_bytes.argtypes = [C.c_int64, C.c_int, C.c_int]
    seg_args = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.hmm_cosine_topk_segmented.restype = C.c_int
    lib.hmm_cosine_topk_segmented.argtypes = seg_args
    lib.hmm_cosine_topk_segmented_prefilter.restype = C.c_int
    lib.hmm_cosine_topk_segmented_prefilter.argtypes = [C.c_void_p, C.c_void_p] + seg_args[1:]

N, K = 1_000_000, 32
g = torch.Generator(device="cuda").manual_seed(42)
rows = torch.empty(N, 1024, device="cuda")
for s in range(0, N, 125_000):
    blk = torch.randn(125_000, 1024, generator=g, device="cuda")
    rows[s:s + 125_000] = blk / blk.norm(dim=1, keepdim=True)
q = torch.randn(1024, generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
store = FeatureStore(rows)
store.build_shadow()
cand = torch.empty(2048 * K, dtype=torch.int64, device="cuda")
ws = torch.empty(max(new.hmm_cosine_topk_prefilter_workspace_bytes(N, K), old.hmm_cosine_topk_prefilter_workspace_bytes(N, K)) + 4096, dtype=torch.uint8, device="cuda")
idx = torch.empty(K, dtype=torch.int64, device="cuda")
sims = torch.empty(K, dtype=torch.float32, device="cuda")
n_out = torch.empty(1, dtype=torch.int32, device="cuda")
E, KS = 2000, 5
offsets = torch.arange(0, N + 1, N // E, dtype=torch.int64, device="cuda")
seg_idx = torch.empty(E, KS, dtype=torch.int64, device="cuda")
seg_sims = torch.empty(E, KS, dtype=torch.float32, device="cuda")
seg_n = torch.empty(E, dtype=torch.int32, device="cuda")
ws_seg = torch.empty(max(new.hmm_cosine_topk_segmented_workspace_bytes(N, E, KS), old.hmm_cosine_topk_segmented_workspace_bytes(N, E, KS)) + 4096,
                     dtype=torch.uint8, device="cuda")


def timed(fn, iters=60):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def legs(lib):
    sp = torch.cuda.current_stream().cuda_stream
    return {"scan_kernel_only": lambda: lib.hmm_op_scan_topk_only(rows.data_ptr(), N, q.data_ptr(), K, cand.data_ptr(), sp),
            "exact_query": lambda: lib.hmm_cosine_topk(rows.data_ptr(), N, 1024, q.data_ptr(), K, idx.data_ptr(), sims.data_ptr(), n_out.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), sp),
            "prefilter_query": lambda: lib.hmm_cosine_topk_prefilter(rows.data_ptr(), store._shadow.data_ptr(), N, 1024, q.data_ptr(), K, idx.data_ptr(),
                                                                     sims.data_ptr(), n_out.data_ptr(), None, ws.data_ptr(), ws.numel(), sp),
            "per_event_2000": lambda: lib.hmm_cosine_topk_segmented(rows.data_ptr(), N, 1024, q.data_ptr(), offsets.data_ptr(), E, KS, seg_idx.data_ptr(),
                                                                    seg_sims.data_ptr(), seg_n.data_ptr(), ws_seg.data_ptr(), ws_seg.numel(), sp),
            "per_event_2000_prefilter": lambda: lib.hmm_cosine_topk_segmented_prefilter(
                rows.data_ptr(), store._shadow.data_ptr(), N, 1024, q.data_ptr(), offsets.data_ptr(), E, KS, seg_idx.data_ptr(), seg_sims.data_ptr(),
                seg_n.data_ptr(), ws_seg.data_ptr(), ws_seg.numel(), sp)}


out = {}
res = {}
for rep in range(3):
    for tag, lib in (("round5", old), ("round6", new), *extra.items()):
        for name, fn in legs(lib).items():
            assert fn() == 0
            out.setdefault(f"{name}_{tag}_ms", []).append(round(timed(fn), 4))
            if name.endswith("_query"):
                res[(name, tag)] = (idx.clone(), sims.clone())
            elif name.startswith("per_event"):
                res[(name, tag)] = (seg_idx.clone(), seg_sims.clone())
summary = {k: min(v) for k, v in out.items()}
summary["same_results"] = all(torch.equal(res[(n, "round5")][0], res[(n, "round6")][0]) and
                              torch.equal(res[(n, "round5")][1].view(torch.int32), res[(n, "round6")][1].view(torch.int32))
                              for n in ("exact_query", "prefilter_query", "per_event_2000", "per_event_2000_prefilter"))
print(json.dumps({"best": summary}, indent=1))
outs = [a for a in sys.argv[1:] if a.endswith(".json")]
if outs:
    json.dump({"all": out, "best": summary}, open(outs[0], "w"), indent=1)
