"""Does running sub-batches on several HIP streams overlap HBM-bound and MFMA-bound kernels?"""
import sys
sys.path.insert(0, ".")
import torch
from hippomm_amd.encoder import HipTower, synthetic_state_dict
sd = synthetic_state_dict(("vision",))
towers = [HipTower("vision", sd) for _ in range(4)]
del sd
B = 256
x = torch.randn(B, 3, 224, 224, device="cuda")
out = torch.empty(B, 1024, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]

def single():
    towers[0].forward_into(x, out)

def multi(n, offset_cycles=0):
    cur = torch.cuda.current_stream()
    per = B // n
    for i in range(n):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            if offset_cycles and i:
                torch.cuda._sleep(int(offset_cycles * i))
            towers[i].forward_into(x[i * per:(i + 1) * per], out[i * per:(i + 1) * per])
    for i in range(n):
        cur.wait_stream(streams[i])

def timeit(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

print(f"single B=256: {timeit(single):.2f} ms", flush=True)
for off_ms in (0, 0.3, 0.7, 1.1, 1.5, 2.0):
    cyc = off_ms * 1.29e6            # torch.cuda._sleep(100000) measured 0.0777 ms
    print(f"2 streams offset {off_ms} ms(nominal): {timeit(lambda: multi(2, cyc)):.2f} ms", flush=True)
print(f"4 streams: {timeit(lambda: multi(4)):.2f} ms", flush=True)
print(f"single again: {timeit(single):.2f} ms", flush=True)
# calibrate _sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(100000); e1.record(); torch.cuda.synchronize()
print("sleep(100000) =", e0.elapsed_time(e1), "ms")
