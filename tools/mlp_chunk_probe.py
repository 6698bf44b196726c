"""Round-4 verdict item 6, the one un-tried lever that removes joules rather than cycles from the batch-256 forward: keep the MLP
hidden activation (R x 5120 bf16 = 674 MB per 256 frames) out of HBM by running fc1 -> fc2 per chunk of <= 16 K token rows with
every chunk's hidden in the SAME buffer (<= 168 MB: inside the 256-MB Infinity Cache).  Forward time and board power (this GPU's
hwmon sensor) for one / two chains x chunk sizes, interleaved twice.  Kill criterion: keep only if the forward gains >= 2 %.
usage: mlp_chunk_probe.py [out.json]"""
import json
import sys
import threading
import time

import torch

from probe_common import load_probe, own_power_file, setter

L, lib = load_probe()
set_chunk = setter(lib, "g_enc_mlp_chunk_rows")
from hippomm_amd.encoder import HipTower, synthetic_state_dict  # noqa: E402

PFILE = own_power_file()
samples, stop = [], False


def sampler():
    while not stop:
        try:
            samples.append((time.perf_counter(), int(open(PFILE).read()) / 1e6))
        except (OSError, ValueError):
            pass
        time.sleep(0.01)


def watts(t0, t1):
    v = [w for t, w in samples if t0 + 0.3 <= t <= t1 - 0.05]
    return sum(v) / len(v) if v else float("nan")


def loop(fn, secs, batch=4):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(batch):
            fn()
        torch.cuda.synchronize(); n += batch
    t1 = time.perf_counter()
    return (t1 - t0) / n * 1e3, watts(t0, t1)


if PFILE:
    th = threading.Thread(target=sampler); th.start()
B = 256
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
frames = torch.randn(B, 3, 224, 224, device="cuda"); emb = torch.empty(B, 1024, device="cuda")
rows, base = [], None
for rep in range(2):
    for streams, chunk in ((2, 0), (1, 0), (1, 16448), (1, 8224), (1, 32896), (2, 8224), (2, 16448), (2, 4112)):
        tower.set_streams(streams)
        set_chunk(chunk)
        ms, w = loop(lambda: tower.forward_into(frames, emb), 3.0)
        out = emb.clone()
        if base is None:
            base = out
        rec = {"rep": rep, "chains": streams, "mlp_chunk_rows": chunk, "hidden_MB_per_chain": round((chunk or (B // streams) * 257) * 5120 * 2 / 1e6),
               "ms": round(ms, 3), "frames_per_s": round(B / ms * 1e3, 1), "board_W": round(w, 1) if PFILE else None,
               "J_per_frame": round(ms * w / 1e3 / B, 4) if PFILE else None, "same_bits": bool(torch.equal(out, base))}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
set_chunk(0)
stop = True
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
