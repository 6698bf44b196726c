"""Is the chip at its power cap while the GEMMs run?  Samples hwmon power / sclk from sysfs in a thread while a kernel loops.
usage: power_probe.py"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hippomm_amd import _lib as L

lib = L.load()


def read(path):
    try:
        return open(path).read().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"


hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon dirs:", hw)
for h in hw[:2]:
    for f in ("power1_cap", "power1_cap_max", "power1_average", "power1_input", "freq1_input", "temp1_input"):
        print(h, f, read(os.path.join(h, f)))
for c in sorted(glob.glob("/sys/class/drm/card*/device"))[:2]:
    print(c, "pp_dpm_sclk:", read(os.path.join(c, "pp_dpm_sclk")).replace("\n", " | "))

samples, stop = [], False


def sampler():
    while not stop:
        row = [time.perf_counter()]
        for h in hw:
            row += [read(os.path.join(h, "power1_input")), read(os.path.join(h, "freq1_input"))]
        samples.append(row)
        time.sleep(0.02)


M = 65792
def gemm_loop(name, N, K, epi, secs):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, L.stream_ptr()), "gemm")
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    print(f"{name}: {2*M*N*K*n/dt/1e12:.0f} TFLOP/s over {dt:.1f} s", flush=True)


t = threading.Thread(target=sampler); t.start()
time.sleep(0.5)
marks = [("idle", time.perf_counter())]
gemm_loop("fc2", 1280, 5120, 2, 3.0); marks.append(("fc2", time.perf_counter()))
time.sleep(0.5); marks.append(("idle2", time.perf_counter()))
gemm_loop("qkv", 3840, 1280, 0, 3.0); marks.append(("qkv", time.perf_counter()))
a = torch.randn(65792, 1280, device="cuda"); y = torch.empty(65792, 1280, dtype=torch.bfloat16, device="cuda")
g = torch.ones(1280, device="cuda"); b = torch.zeros(1280, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    for _ in range(50):
        lib.hmm_op_layernorm_bf16(a.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), 65792, 1280, 1e-6, L.stream_ptr())
    torch.cuda.synchronize()
marks.append(("layernorm", time.perf_counter()))
del a, y
# the whole ViT-H forward at 256 frames: two half-batch chains (the shipped default) and one chain
from hippomm_amd.encoder import HipTower, synthetic_state_dict
tower = HipTower("vision", synthetic_state_dict(("vision",), seed=1234))
frames = torch.randn(256, 3, 224, 224, device="cuda"); emb = torch.empty(256, 1024, device="cuda")
for streams in (2, 1):
    tower.set_streams(streams)
    time.sleep(0.5); marks.append((f"idle{streams}", time.perf_counter()))
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(5):
            tower.forward_into(frames, emb)
        torch.cuda.synchronize(); n += 5
    dt = time.perf_counter() - t0
    print(f"forward, {streams} stream(s): {dt / n * 1e3:.2f} ms per 256 frames", flush=True)
    marks.append((f"forward_{streams}s", time.perf_counter()))
stop = True; t.join()
prev = samples[0][0]
for name, tm in marks:
    seg = [s for s in samples if prev <= s[0] < tm]
    prev = tm
    def avg(i):
        v = [float(s[i]) for s in seg if s[i].replace(".", "").isdigit()]
        return round(sum(v) / len(v) / 1e6) if v else None
    def mx(i):
        v = [float(s[i]) for s in seg if s[i].replace(".", "").isdigit()]
        return round(max(v) / 1e6) if v else None
    print(f"{name:10s} samples {len(seg):4d}  " + "  ".join(f"[{k}] {avg(1 + 2 * k)} W (max {mx(1 + 2 * k)}) {avg(2 + 2 * k)} MHz" for k in range(len(hw))))
