"""GPU: bench.py's N > 1 code path, rehearsed on one GPU (HMM_BENCH_REHEARSAL=1: two ranks, both on cuda:0, collectives over
gloo).  Everything the driver's multi-GPU run executes except RCCL itself: self-launch through torch.distributed.run,
frame sharding, the embedding all-gather, selection on the gathered matrix, the row-sharded scan with its key exchange and
merge, rank-0 JSON line.  The numbers are meaningless (two ranks share a GPU) and the line says so."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_two_rank_rehearsal_prints_a_complete_line():
    env = dict(os.environ, HMM_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--scan-strong"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "frame-embeddings/s" and d["value"] > 0
    assert d["config"]["of_frames"] == 512 and d["config"]["sharding"] == "frames x2" and "all_gather_ms" in d["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    scan = d["scan"]
    assert "error" not in scan, scan
    for tag, rows in (("weak_1M_rows_per_gpu", 1_000_000), ("strong_1M_rows_total", 500_000)):
        assert scan[tag]["rows_per_gpu"] == rows and scan[tag]["indices_match_torch_where_separated"] is True
    assert "rehearsal" in d                                  # never mistaken for a measurement
