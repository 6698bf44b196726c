"""GPU, boxes with >= 2 devices only (skipped on the 1-GPU boxes of this pool): first-contact insurance for what no test here has
ever run on hardware -- RCCL above world size 1 and the library on a device other than 0.

* BASELINE cfg 5 through bench.py on two GPUs over the REAL RCCL backend: contiguous time shards -> encode -> one
  all_gather_into_tensor over xGMI -> selection on every rank; gathered-matrix and kept-list SHA-256 equal to the one-GPU run
  (the same asserts as the gloo rehearsals in tests/test_gpu_configs.py).
* Two devices in one process: tower, scan, prefilter, per-event scan and selection on cuda:1 while cuda:0 is the default device
  -- the per-(kernel, device) dynamic-LDS attribute (hmm_common.h HMM_ENSURE_DYN_LDS) and the handle-device check."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import imagebind_oracle as ib
from oracle.consolidation_oracle import select_key_frames_oracle
from oracle.vector_ops_oracle import top_k_cosine_similarity_oracle

ROOT = Path(__file__).resolve().parent.parent
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs (device_count() does not open the runtime)")]


def _bench(args, env=None):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=1700,
                       env=dict(os.environ, **(env or {})), cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_cfg5_two_gpus_over_rccl_equals_one_gpu_bitwise():
    common = ["--workload", "cfg5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = _bench(common + ["--gpus", "1"])
    two = _bench(common + ["--gpus", "2", "--no-scan"])
    c, c1 = two["config"], one["config"]
    assert two["n_gpus"] == 2 and "rehearsal" not in two
    assert two["collective_backend"].startswith("nccl") and two["rccl_ranks"] == 2 and two["all_reduced_rank_count"] == 2
    assert len({r["pci_bus_id"] for r in two["ranks"]}) == 2, "both ranks report the same GPU"
    assert c["frames_per_rank"] == [1800, 1800] and c["all_gather_ms"] > 0
    assert c["kept_equal_cpu_oracle_on_gathered_matrix"] is True
    assert c["kept_indices_sha256"] == c1["kept_indices_sha256"]
    assert c["gathered_embeddings_sha256"] == c1["gathered_embeddings_sha256"]


def test_cfg2_two_gpus_with_the_sharded_scan_leg():
    d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["rccl_ranks"] == 2
    assert d["config"]["frames_per_gpu"] == 256 and d["value"] > 0
    assert d["scan"]["weak_1M_rows_per_gpu"]["indices_match_torch_where_separated"] is True


def test_library_on_the_second_device_of_one_process():
    from hippomm_amd import _lib as L
    from hippomm_amd.consolidation import select_key_frames_device
    from hippomm_amd.encoder import HipTower
    from hippomm_amd.vector_ops import EventStore, FeatureStore
    torch.cuda.set_device(0)
    dev1 = torch.device("cuda", 1)
    spec = ib.reduced(ib.VISION_HUGE, 2)
    st = ib.synthetic_state(spec, seed=21, init="rich")
    x = torch.randn(50, 3, 224, 224, generator=torch.Generator().manual_seed(4))
    t0 = HipTower("vision", st, depth=2)                       # device 0 (current)
    t1 = HipTower("vision", st, depth=2, device=dev1)          # device 1: its own LDS attributes, streams, weights
    e0, e1 = t0(x), t1(x)
    assert e1.device == dev1 and torch.equal(e0.cpu(), e1.cpu())          # 50 frames: two chains + the fused kernel (dynamic LDS)
    assert torch.equal(t1(x[:1]).cpu(), t0(x[:1]).cpu())                  # few-row regime: split-K rings on device 1
    want = ib.vision_forward(x[:2], st, spec)
    assert (1 - torch.nn.functional.cosine_similarity(e1[:2].cpu(), want)).max().item() <= 5e-5
    # a forward issued while another device is current is refused, not silently run with the wrong streams
    lib = L.load()
    out = torch.empty(1, 1024, device=dev1)
    xx = x[:1].to(dev1)
    ws = torch.empty(lib.hmm_encoder_workspace_bytes(t1._h, 1), dtype=torch.uint8, device=dev1)
    rc = lib.hmm_encoder_forward(t1._h, xx.data_ptr(), 1, out.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr())
    assert rc == -4 and b"current device" in lib.hmm_last_error()
    # scan, prefilter, per-event scan and selection on device 1
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((40000, 1024), dtype=np.float32)
    q = rng.standard_normal(1024, dtype=np.float32)
    with torch.cuda.device(dev1):
        fs = FeatureStore(rows, device=dev1)
        idx, sims = fs.search(q, 9)
        fs.build_shadow()
        i2, s2 = fs.search_prefiltered_device(torch.from_numpy(q).to(dev1), 9)
        ev = EventStore([rows[:25000], rows[25000:]], device=dev1)
        per = ev.top_k_per_event(q, 5)
        m = ev.search_multi(np.stack([q, -q]), 4)
        feats = torch.from_numpy(rows[:300] + 3.0).to(dev1)
        kept = select_key_frames_device(feats).cpu().numpy()
    w_idx, w_sims = top_k_cosine_similarity_oracle(q, rows, 9)
    assert idx.tolist() == w_idx.tolist() and np.allclose(sims, w_sims, atol=2e-6)
    assert i2.cpu().tolist() == idx.tolist() and np.array_equal(s2.cpu().numpy(), sims)
    assert per[1][0].tolist() == top_k_cosine_similarity_oracle(q, rows[25000:], 5)[0].tolist()
    assert m[1][0].tolist() == top_k_cosine_similarity_oracle(-q, rows, 4)[0].tolist()
    assert kept.tolist() == select_key_frames_oracle(rows[:300] + 3.0).tolist()
