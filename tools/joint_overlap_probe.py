"""BASELINE cfg 3 (128 frame + audio-segment pairs): the two towers one after the other on one stream (bench.py's joint_bench) against
the two towers launched on two streams (they are independent), interleaved; and single-chain towers on two streams.
usage: joint_overlap_probe.py [out.json]"""
import json
import sys

import torch

from probe_common import ROOT  # noqa: F401
import bench
from hippomm_amd.encoder import HipTower, synthetic_state_dict

pairs = 128
sd = synthetic_state_dict(("vision", "audio"), seed=1234)
vis, aud = HipTower("vision", sd), HipTower("audio", sd)
del sd
frames = torch.randn(pairs, 3, 224, 224, device="cuda")
mels = torch.randn(pairs, 3, 1, 128, 204, device="cuda")
ev, ea = torch.empty(pairs, 1024, device="cuda"), torch.empty(pairs, 1024, device="cuda")
vis._workspace(pairs); aud._workspace(pairs)
s2 = torch.cuda.Stream()


def sequential():
    vis.forward_into(frames, ev)
    aud.forward_into(mels, ea)


def overlapped():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    vis.forward_into(frames, ev)
    with torch.cuda.stream(s2):
        aud.forward_into(mels, ea)
    cur.wait_stream(s2)


rows = []
ref_v, ref_a = None, None
for rep in range(3):
    for tag, fn, streams in (("sequential (bench.py)", sequential, 2), ("two streams", overlapped, 2), ("two streams, one chain per tower", overlapped, 1),
                             ("sequential, one chain per tower", sequential, 1)):
        vis.set_streams(streams); aud.set_streams(streams)
        ms = bench.event_time_ms(fn, 8, warmup=3)
        if ref_v is None:
            ref_v, ref_a = ev.clone(), ea.clone()
        rec = {"rep": rep, "schedule": tag, "ms": round(ms, 3), "pairs_per_s": round(pairs / ms * 1e3, 1),
               "same_bits": bool(torch.equal(ev, ref_v) and torch.equal(ea, ref_a))}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
