"""A few forwards of one tower at one batch size, for `rocprofv3 --kernel-trace`: python3 tools/trace_forward.py vision 32
[one_chain]: with a third argument the tower runs as ONE chain (hmm_encoder_set_streams(1)) whatever its size."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from hippomm_amd.encoder import HipTower, synthetic_state_dict
kind, B = sys.argv[1], int(sys.argv[2])
tower = HipTower(kind, synthetic_state_dict((kind,), seed=1))
if kind == "text":
    x = torch.randint(1, 49000, (B, 77), device="cuda"); x[:, 0], x[:, 20] = 49406, 49407
else:
    x = torch.randn(B, 3, 224, 224, device="cuda") if kind == "vision" else torch.randn(B, 3, 1, 128, 204, device="cuda")
out = torch.empty(B, 1024, device="cuda")
if len(sys.argv) > 3:
    tower.set_streams(1)
for _ in range(6):
    tower.forward_into(x, out)
torch.cuda.synchronize()
