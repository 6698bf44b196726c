"""Encoder latency at the reference's call sizes (32-frame buffer, single segments), eager vs HIP graph."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from hippomm_amd.encoder import HipTower, synthetic_state_dict
sd = synthetic_state_dict(("vision",))
tower = HipTower("vision", sd); del sd
for B in (1, 8, 32, 64, 128):
    x = torch.randn(B, 3, 224, 224, device="cuda"); out = torch.empty(B, 1024, device="cuda")
    for _ in range(3): tower.forward_into(x, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): tower.forward_into(x, out)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / 10 * 1e3
    # graph
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tower.forward_into(x, out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            tower.forward_into(x, out)
    torch.cuda.synchronize()
    ref = out.clone()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t) / 10 * 1e3
    print(f"B={B:4d}: eager {eager:7.2f} ms ({B/eager*1e3:7.0f} img/s)   graph {graph:7.2f} ms ({B/graph*1e3:7.0f} img/s)   same={torch.equal(ref, out)}", flush=True)
