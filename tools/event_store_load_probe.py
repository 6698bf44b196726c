"""Cold and warm loads of a synthetic memory_store (events of 600 vision rows written in the reference's JSON format): the reference's
reading rule (json.load + np.array, serial), the library's matrix parser serial and on 16 threads, and the fp32 sidecars; then the
resident EventStore build and one per-event query.  usage: event_store_load_probe.py [events] [out.json]"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                 # noqa: E402
import torch                                                       # noqa: E402
from hippomm_amd import event_store as es                          # noqa: E402

n_events = int(sys.argv[1]) if len(sys.argv) > 1 else 48
base = tempfile.mkdtemp(prefix="hmm_store_")
rng = np.random.default_rng(0)
index = {}
t0 = time.perf_counter()
for i in range(n_events):
    f = rng.standard_normal((600, 1024)).astype(np.float32)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    ev = {"features": {"vision": f, "vision_times": np.arange(600) * 1.0}, "frames": [f"f{j}.jpg" for j in range(40)], "frame_times": list(range(40)),
          "frame_captions": ["c"] * 40, "audio_times": [0.0], "audio_transcription": ["hi"], "holistic_audio_transcription": "hi", "summary": "s",
          "start_time": 0.0, "end_time": 600.0}
    p = es.save_event(ev, os.path.join(base, "events", "vid", f"vid_{i}.json"), write_sidecars=False)
    index[f"vid_{i}"] = {"file_path": str(p), "video_id": "vid"}
write_s = time.perf_counter() - t0
with open(os.path.join(base, "event_index.json"), "w") as fh:
    json.dump(index, fh)
files = [p for _, p in es.iter_event_files(base)]
size_mb = sum(os.path.getsize(p) for p in files) / 1e6
rec = {"events": n_events, "rows": n_events * 600, "json_mb": round(size_mb, 1), "host_threads": os.cpu_count(),
       "write_s_per_event": round(write_s / n_events, 3)}
sample = files[: min(8, n_events)]
t0 = time.perf_counter()
for p in sample:
    es.parse_event_features(p, native=False)
rec["reference_rule_ms_per_event"] = round((time.perf_counter() - t0) / len(sample) * 1e3, 1)
t0 = time.perf_counter()
for p in sample:
    es.parse_event_features(p, native=True)
rec["library_serial_ms_per_event"] = round((time.perf_counter() - t0) / len(sample) * 1e3, 1)
t0 = time.perf_counter()
store, ids = es.build_event_store(base, "vision", workers=16)      # cold: parses every file on 16 threads, writes the sidecars
torch.cuda.synchronize()
rec["cold_build_event_store_s"] = round(time.perf_counter() - t0, 3)
rec["cold_ms_per_event"] = round(rec["cold_build_event_store_s"] / n_events * 1e3, 1)
t0 = time.perf_counter()
store, ids = es.build_event_store(base, "vision", workers=16)      # warm: the sidecars
torch.cuda.synchronize()
rec["warm_build_event_store_s"] = round(time.perf_counter() - t0, 3)
q = torch.randn(1024, device="cuda")
hits = store.top_k_per_event(q, 5)
torch.cuda.synchronize()
rec["reference_rule_whole_store_s_extrapolated"] = round(rec["reference_rule_ms_per_event"] * n_events / 1e3, 1)
print(json.dumps(rec), flush=True)
if len(sys.argv) > 2:
    json.dump(rec, open(sys.argv[2], "w"), indent=1)
shutil.rmtree(base, ignore_errors=True)
