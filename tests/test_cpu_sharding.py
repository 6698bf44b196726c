"""CPU, world_size 2, gloo: the N>1 exchange logic of hippomm_amd.sharding -- ragged all-gather in
time order, global selection == single-process selection, sharded top-k merge on global indices.
The compute legs are stood in by the oracle (the HIP kernels are covered by the -m gpu tests)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recipes
        from hippomm_amd import sharding
        from oracle.consolidation_oracle import select_key_frames_oracle
        from oracle.vector_ops_oracle import scan_order_key

        feats_all = torch.from_numpy(recipes.clustered(n_frames, max(n_frames // 6, 1), 0.2, seed=99))
        bounds = sharding.shard_bounds(n_frames, world)
        lo, hi = bounds[rank]
        counts = [b - a for a, b in bounds]

        # "frames" here are the features themselves; encode_fn = identity stands for the tower
        def select_fn(f, thr):
            return torch.from_numpy(select_key_frames_oracle(f.numpy(), None, thr).astype(np.int64))

        feats, kept = sharding.consolidate_sharded(feats_all[lo:hi], lambda x: x, counts, 0.9, select_fn)
        ok_gather = torch.equal(feats, feats_all)
        want = select_key_frames_oracle(feats_all.numpy(), None, 0.9)
        ok_kept = kept.tolist() == want.tolist()
        # without counts (size exchange path)
        feats2 = sharding.all_gather_embeddings(feats_all[lo:hi])
        ok_gather2 = torch.equal(feats2, feats_all)
        # from FILES: every rank gets the whole list of paths, embeds its own contiguous shard ("path" i stands for frame i)
        paths = [f"frame_{i:05d}.jpg" for i in range(n_frames)]
        seen = []

        def extract(local_paths):
            seen.extend(local_paths)
            return feats_all[[int(p[6:11]) for p in local_paths]]
        feats3, kept3 = sharding.consolidate_paths_sharded(paths, extract, 0.9, select_fn, device=torch.device("cpu"))
        ok_gather2 = ok_gather2 and torch.equal(feats3, feats_all) and kept3.tolist() == want.tolist() and seen == paths[lo:hi]

        # sharded top-k over the same rows: local keys from the oracle's order key, merge on CPU
        query = torch.from_numpy(np.random.default_rng(5).standard_normal(1024).astype(np.float32))
        store = feats_all.numpy().copy()
        store[min(3, n_frames - 1)] = store[n_frames - 1]          # a tie across shards
        k = 6

        def sims_of(rows):
            with np.errstate(invalid="ignore", divide="ignore"):
                return ((rows @ query.numpy()) / (np.linalg.norm(rows, axis=1) * np.linalg.norm(query.numpy()))).astype(np.float32)

        def local_keys(qv, kk):
            key = np.sort(scan_order_key(sims_of(store[lo:hi])))[::-1][:kk]
            out = np.zeros(kk, dtype=np.uint64)
            out[: len(key)] = key
            return torch.from_numpy(out.view(np.int64).copy())

        def merge(keys, offs, kk):
            ks = keys.numpy().view(np.uint64).reshape(-1, kk)
            glob = []
            for s in range(ks.shape[0]):
                for key in ks[s]:
                    if key:
                        glob.append((int(key) & ~0xFFFFFFFF) | ((int(key) & 0xFFFFFFFF) + int(offs[s])))
            glob = sorted(glob, reverse=True)[:kk]
            return torch.tensor([g & 0xFFFFFFFF for g in glob]), None

        idx, _ = sharding.sharded_top_k(query, k, hi - lo, lo, local_keys, merge)
        full = np.sort(scan_order_key(sims_of(store)))[::-1][:k]
        ok_topk = idx.tolist() == [int(x) & 0xFFFFFFFF for x in full]
        q.put((rank, ok_gather, ok_kept, ok_gather2, ok_topk))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [37, 64, 3])
def test_world2_gloo(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_frames) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, *oks in results:
        assert all(oks), f"rank {rank}: gather/kept/gather2/topk = {oks}"


def test_shard_bounds_cover_in_time_order():
    from hippomm_amd.sharding import shard_bounds
    for n, w in [(3600, 8), (37, 2), (3, 8), (0, 4), (256, 1)]:
        b = shard_bounds(n, w)
        assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(e - s for s, e in b) == (n + w - 1) // w if n else True
