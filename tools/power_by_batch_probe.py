"""Board power (this GPU's hwmon sensor) and shader clock while the vision forward loops at 1 ... 256 frames: from which batch size
on does the forward run at the board's power cap (LABNOTES 4.9), i.e. where would a shorter kernel stop buying time?
usage: power_by_batch_probe.py [json_out]"""
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from probe_common import own_power_file                           # noqa: E402
import torch                                                       # noqa: E402
from hippomm_amd.encoder import HipTower, synthetic_state_dict    # noqa: E402

PFILE = own_power_file()
if PFILE is None:
    raise SystemExit("no hwmon power sensor for this GPU")
FFILES = glob.glob(os.path.join(os.path.dirname(PFILE), "freq1_input"))
CAPF = os.path.join(os.path.dirname(PFILE), "power1_cap")
samples, stop = [], False


def sampler():
    while not stop:
        try:
            f = int(open(FFILES[0]).read()) / 1e6 if FFILES else float("nan")
            samples.append((time.perf_counter(), int(open(PFILE).read()) / 1e6, f))
        except (OSError, ValueError):
            pass
        time.sleep(0.005)


th = threading.Thread(target=sampler, daemon=True)
th.start()
torch.cuda.init()
torch.zeros(1, device="cuda")
rows = []
cap = int(open(CAPF).read()) / 1e6 if os.path.exists(CAPF) else None
time.sleep(3.0)                                                    # the idle board (context created, nothing running)
v = [(w, f) for t, w, f in samples[len(samples) // 3:]]
rows.append({"frames": 0, "what": "idle, HIP context created", "watts_mean": round(sum(w for w, _ in v) / len(v), 1),
             "sclk_mhz_mean": round(sum(f for _, f in v) / len(v), 1), "power_cap_w": cap, "samples": len(v)})
print(json.dumps(rows[-1]), flush=True)
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
for B in (1, 4, 8, 16, 32, 64, 128, 256):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    out = torch.empty(B, 1024, device="cuda")
    for _ in range(5):
        tower.forward_into(x, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(4):
            tower.forward_into(x, out)
        torch.cuda.synchronize()
        n += 4
    t1 = time.perf_counter()
    v = [(w, f) for t, w, f in samples if t0 + 0.5 <= t <= t1 - 0.05]
    rec = {"frames": B, "ms_per_forward": round((t1 - t0) / n * 1e3, 3), "frames_per_s": round(B * n / (t1 - t0), 1),
           "watts_mean": round(sum(w for w, _ in v) / len(v), 1), "watts_max": round(max(w for w, _ in v), 1),
           "sclk_mhz_mean": round(sum(f for _, f in v) / len(v), 1), "power_cap_w": cap, "samples": len(v)}
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    time.sleep(1.0)
stop = True
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
