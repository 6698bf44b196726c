"""One-process-per-GPU sharding of the hot path over the GPUs of a node.

The reference is single-process (foundation_models.py:26, :33); this is new.  Frames are
independent units, so the encoder shards with no data-path collective:

  * ``shard_bounds``            contiguous, time-ordered shards (row order after concatenation must be
                                time order, which the greedy consolidation rule depends on);
  * ``all_gather_embeddings``   ONE all-gather (RCCL over xGMI; 1.84 MB/rank at n=3600) that reassembles
                                the (n,1024) fp32 embedding matrix on every rank; ragged shards are
                                padded to the largest shard and trimmed after the gather;
  * ``consolidate_sharded``     encode local frames -> all-gather -> global key-frame selection
                                (every rank runs the 26.5 GFLOP selection redundantly: it is cheaper than
                                broadcasting the result and keeps ranks symmetric);
  * ``consolidate_paths_sharded``  the same from frame FILES: every rank runs extract_features on its shard of paths;
  * ``sharded_top_k``           row-sharded feature_search: local scan -> all-gather of k packed
                                (sim,row) keys (8*k bytes/rank) -> merge under the same total order
                                as the single-GPU scan, applied to GLOBAL row indices.

``encode_fn`` / ``select_fn`` / ``local_keys_fn`` default to the HIP path; the CPU (gloo) tests inject
stand-ins to exercise the exchange logic without a GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous shards of ceil(n/world) rows (the last ones may be shorter or empty)."""
    per = (n + world_size - 1) // world_size if n > 0 else 0
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(world_size)]


def all_gather_embeddings(local: torch.Tensor, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """local: (n_local, 1024) on this rank's device -> (sum n_local, 1024) on every rank, rank order.

    ``counts`` (rows per rank) avoids an extra size exchange when the caller sharded with
    ``shard_bounds``; otherwise sizes are gathered first."""
    rank, ws = world()
    if ws == 1:
        return local
    if counts is None:
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        sizes = [torch.zeros_like(mine) for _ in range(ws)]
        dist.all_gather(sizes, mine)
        counts = [int(s.item()) for s in sizes]
    cap = max(counts)
    if cap == 0:
        return local
    if all(c == cap for c in counts):
        out = torch.empty(ws * cap, local.shape[1], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    padded = torch.zeros(cap, local.shape[1], dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty(ws * cap, local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * cap: r * cap + counts[r]] for r in range(ws)], dim=0)


def _hip_select(features: torch.Tensor, thr: float) -> torch.Tensor:
    from .consolidation import select_key_frames_device
    if features.shape[0] <= 2:
        return torch.arange(features.shape[0], dtype=torch.int64, device=features.device)
    return select_key_frames_device(features, thr)


def consolidate_sharded(frames_local: torch.Tensor,
                        encode_fn: Callable[[torch.Tensor], torch.Tensor],
                        counts: Optional[Sequence[int]] = None,
                        similarity_threshold: float = 0.9,
                        select_fn: Optional[Callable[[torch.Tensor, float], torch.Tensor]] = None):
    """BASELINE cfg 5: each rank encodes its contiguous time shard, one all-gather reassembles the
    embedding matrix, every rank selects key frames on the full matrix.

    Returns (features (n,1024) for ALL frames, kept global indices int64) -- the pair
    ``_process_vision_features`` stores (hippocampal_memory.py:858-866)."""
    local = encode_fn(frames_local) if frames_local.shape[0] > 0 else \
        torch.zeros(0, 1024, dtype=torch.float32, device=frames_local.device)
    feats = all_gather_embeddings(local, counts)
    kept = (select_fn or _hip_select)(feats, similarity_threshold)
    return feats, kept


def consolidate_paths_sharded(frame_paths: Sequence[str],
                              extract_fn: Callable[[Sequence[str]], torch.Tensor],
                              similarity_threshold: float = 0.9,
                              select_fn: Optional[Callable[[torch.Tensor, float], torch.Tensor]] = None,
                              device: Optional[torch.device] = None):
    """BASELINE cfg 5 from FILES: the time-ordered frame paths of a video are cut into contiguous shards, every rank runs the
    reference's own call on its shard -- ``extract_fn(paths) = imagebind.extract_features({'vision': paths}, ['vision'])['vision']``
    (hippocampal_memory.py:1180-1186), i.e. decode | upload + resize | tower on that rank's GPU and CPU share -- then ONE all-gather
    and the global selection on every rank.  Returns (features (n,1024) of ALL frames in time order, kept global indices)."""
    rank, ws = world()
    bounds = shard_bounds(len(frame_paths), ws)
    lo, hi = bounds[rank]
    if hi > lo:
        local = extract_fn(list(frame_paths[lo:hi]))
    else:
        local = torch.zeros(0, 1024, dtype=torch.float32, device=device or torch.device("cuda", torch.cuda.current_device()))
    feats = all_gather_embeddings(local, [b - a for a, b in bounds])
    kept = (select_fn or _hip_select)(feats, similarity_threshold)
    return feats, kept


def sharded_top_k(query: torch.Tensor, k: int, n_local: int, row_offset: int,
                  local_keys_fn: Callable[[torch.Tensor, int], torch.Tensor],
                  merge_fn: Optional[Callable] = None):
    """Row-sharded feature_search.  ``local_keys_fn(query, k)`` returns this rank's k packed order
    keys (int64 bit patterns, 0-padded; FeatureStore.search_keys_device).  Every rank returns the
    same (global indices int64[k'], sims fp32[k'])."""
    rank, ws = world()
    keys = local_keys_fn(query, k) if n_local > 0 else torch.zeros(k, dtype=torch.int64, device=query.device)
    offs = torch.tensor([row_offset], dtype=torch.int64, device=query.device)
    if ws > 1:
        all_keys = torch.empty(ws * k, dtype=torch.int64, device=query.device)
        dist.all_gather_into_tensor(all_keys, keys.contiguous())
        all_offs = torch.empty(ws, dtype=torch.int64, device=query.device)
        dist.all_gather_into_tensor(all_offs, offs)
    else:
        all_keys, all_offs = keys, offs
    if merge_fn is None:
        from .vector_ops import merge_keys_device
        merge_fn = merge_keys_device
    return merge_fn(all_keys.reshape(ws, k), all_offs, k)
