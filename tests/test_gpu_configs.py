"""GPU: BASELINE.json's configs run as WORKLOADS, end to end.

cfg 5 (1-hour video at 1 fps: 3600 frames in contiguous time shards -> encode -> ONE all-gather -> global key-frame
selection; reference path being sharded: hippomm/core/hippocampal_memory.py:1172-1186 -> :815-867 -> :944-967) through
bench.py itself, at N = 1 and as a two-rank rehearsal on one GPU (both ranks on cuda:0, collectives over gloo: launch,
time shards, gather and selection execute as on a node).  The kept indices must equal the CPU oracle's on the gathered
matrix, and the gathered matrix and the kept list must be BITWISE the same at both N (time shards + batch invariance of
the tower).

cfg 1 (the reference's own 32-frame buffer: _process_frame_batch, hippocampal_memory.py:1328-1335 -> :855 -> :944-967) as
one chain at full depth: 32 frames -> ImageBind.extract_features -> _select_key_frames, embeddings against the fp32 oracle
tower, kept indices against the selection oracle fed the same embeddings."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(args, env=None):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, **(env or {})), cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def cfg5_one_gpu():
    return _bench(["--workload", "cfg5", "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])


def test_cfg5_on_one_gpu(cfg5_one_gpu):
    d = cfg5_one_gpu
    c = d["config"]
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["unit"] == "frame-embeddings/s" and d["value"] > 0
    assert c["frames_total"] == 3600 and c["frames_per_rank"] == [3600] and "cfg5" in c["workload"]
    assert c["kept_equal_cpu_oracle_on_gathered_matrix"] is True
    assert 1 <= c["kept_key_frames"] < 3600                 # near-duplicate frames: the selection really drops frames
    assert abs(d["value"] - 3600 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1.0
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])


def test_cfg5_two_rank_rehearsal_reproduces_the_one_gpu_result_bitwise(cfg5_one_gpu):
    d = _bench(["--workload", "cfg5", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-scan"],
               env={"HMM_BENCH_REHEARSAL": "1"})
    c, c1 = d["config"], cfg5_one_gpu["config"]
    assert d["n_gpus"] == 2 and "rehearsal" in d and c["frames_per_rank"] == [1800, 1800] and "all_gather_ms" in c
    assert d["all_reduced_rank_count"] == 2 and [r["frames"] for r in d["ranks"]] == [1800, 1800]
    assert d["collective_backend"] == "gloo" and d["rccl_ranks"] is None        # a rehearsal never claims RCCL
    assert c["kept_equal_cpu_oracle_on_gathered_matrix"] is True
    assert c["kept_key_frames"] == c1["kept_key_frames"]
    assert c["kept_indices_sha256"] == c1["kept_indices_sha256"]
    assert c["gathered_embeddings_sha256"] == c1["gathered_embeddings_sha256"]  # 7 x 256 + 8 per rank vs 14 x 256 + 16


def test_cfg5_eight_rank_rehearsal_reproduces_the_one_gpu_result_bitwise(cfg5_one_gpu):
    """The shape the driver's 8-GPU run has: 8 ranks x 450 frames (256 + 194 per rank through the tower), one all-gather of
    (450,1024) per rank, selection on (3600,1024) on every rank.  Eight processes on one GPU over gloo."""
    d = _bench(["--workload", "cfg5", "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-scan"],
               env={"HMM_BENCH_REHEARSAL": "1"})
    c, c1 = d["config"], cfg5_one_gpu["config"]
    assert d["n_gpus"] == 8 and "rehearsal" in d and c["frames_per_rank"] == [450] * 8 and c["all_gather_bytes_per_rank"] == 450 * 4096
    assert d["all_reduced_rank_count"] == 8 and [r["rank"] for r in d["ranks"]] == list(range(8))
    assert c["kept_equal_cpu_oracle_on_gathered_matrix"] is True
    assert c["kept_indices_sha256"] == c1["kept_indices_sha256"]
    assert c["gathered_embeddings_sha256"] == c1["gathered_embeddings_sha256"]
    print(f"8-rank rehearsal: startup {d['startup_s']} s on rank 0, step {d['ms_per_step']} ms (eight ranks sharing one GPU)")


def test_cfg5_ragged_shards_eight_ranks_equal_one_gpu_bitwise():
    """3601 frames do not divide by 8: shard_bounds gives 7 x 451 + 444, the all-gather pads to 451 rows and trims."""
    one = _bench(["--workload", "cfg5", "--cfg5-frames", "3601", "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    d = _bench(["--workload", "cfg5", "--cfg5-frames", "3601", "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                "--no-scan"], env={"HMM_BENCH_REHEARSAL": "1"})
    c, c1 = d["config"], one["config"]
    assert c["frames_per_rank"] == [451] * 7 + [444] and c["all_gather_bytes_per_rank"] == 451 * 4096
    assert c["kept_equal_cpu_oracle_on_gathered_matrix"] is True and c1["kept_equal_cpu_oracle_on_gathered_matrix"] is True
    assert c["kept_indices_sha256"] == c1["kept_indices_sha256"]
    assert c["gathered_embeddings_sha256"] == c1["gathered_embeddings_sha256"]


def test_cfg1_frame_buffer_chain_at_full_depth():
    import bench
    from hippomm_amd import consolidation
    from hippomm_amd.encoder import ImageBind, synthetic_state_dict
    from oracle import imagebind_oracle as ib
    from oracle.consolidation_oracle import evaluated_margin, select_key_frames_oracle

    sd = synthetic_state_dict(("vision",), seed=1234)
    model = ImageBind(state_dict=sd, towers=("vision",))
    frames = bench.synthetic_frames(0, 32, "cuda")                           # processing.frame_buffer_size = 32
    feats = model.extract_features({"vision": frames}, ["vision"])["vision"]
    emb = feats.detach().cpu().numpy()                                       # as hippocampal_memory.py:1335
    assert emb.shape == (32, 1024) and emb.dtype == np.float32

    class Memory:                                                            # stands for HippocampalMemory: self is unused
        _select_key_frames = consolidation._select_key_frames
    times = np.arange(32, dtype=np.float64)
    kept = Memory()._select_key_frames(emb, times)
    want = select_key_frames_oracle(emb, times, 0.9)
    assert evaluated_margin(emb, 0.9) > 1e-6, "a comparison of this input sits inside the sgemm-order band"
    assert kept.dtype == np.int64 and kept.tolist() == want.tolist()
    assert kept[0] == 0 and 1 <= len(kept) < 32, "six scenes of near-duplicates: the selection must drop frames"
    print(f"cfg 1: kept {kept.tolist()} of 32, nearest comparison {evaluated_margin(emb, 0.9):.2e} from the threshold")

    st = {k: v.detach().float().cpu() for k, v in sd.items()}
    ref = ib.vision_forward(frames.cpu(), st)                                # fp32 oracle tower, all 32 blocks
    cos = torch.nn.functional.cosine_similarity(feats.cpu(), ref, dim=1)
    assert (1 - cos).max().item() <= 5e-5, cos.min().item()
    assert (feats.cpu() - ref).abs().max().item() <= 2e-3
    # fed the ORACLE's fp32 embeddings the selection keeps the same frames unless a comparison sits within the bf16
    # embedding error of the threshold (north_star: bit-exact indices are defined on identical feature matrices)
    if min(evaluated_margin(emb, 0.9), evaluated_margin(ref.numpy(), 0.9)) > 1e-3:
        assert select_key_frames_oracle(ref.numpy(), times, 0.9).tolist() == want.tolist()
