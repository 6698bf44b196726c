"""Key-frame selection (hmm_gram_select) timing at the BASELINE sizes, next to the numpy oracle on the host."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hippomm_amd.consolidation import select_key_frames_device
from oracle.consolidation_oracle import select_key_frames_oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import recipes
for name, n in [("n32_clusters6", 32), ("n257_clusters40", 257), ("n3600_clusters600", 3600)]:
    f, t = recipes.select_case(name)
    fd = torch.from_numpy(f).cuda()
    for _ in range(3): kept = select_key_frames_device(fd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): kept = select_key_frames_device(fd)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    want = select_key_frames_oracle(f)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    assert kept.cpu().numpy().tolist() == want.tolist()
    print(f"select n={n}: GPU {gpu_ms:.3f} ms (incl. the count read-back), numpy oracle {cpu_ms:.1f} ms, kept {len(want)}", flush=True)
