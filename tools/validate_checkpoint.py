#!/usr/bin/env python3
"""Validate the towers against a REAL ImageBind checkpoint, when one is supplied.

    python tools/validate_checkpoint.py /path/to/imagebind_huge.pth [--towers vision audio text]
    python tests/checkpoint_vs_oracle.py /path/to/imagebind_huge.pth      # the same + cosine against the fp32 CPU oracle

The encoder oracle (oracle/imagebind_oracle.py) and the HIP towers are restated from the published
architecture with upstream state-dict key names ("parity unpinned": neither the imagebind package nor
imagebind_huge.pth is in this image or in /root/reference; reference foundation_models.py:31-35 downloads it).
This script is the hook that can retire that caveat once a checkpoint is at hand.  It

  1. loads the file with torch.load (CPU) and, per tower, reports which checkpoint keys the tower CONSUMED, which
     keys under the tower's prefixes it left UNUSED (a key the restated architecture does not know = a structural
     mismatch) and which expected keys are MISSING (hmm_encoder_load_param / hmm_encoder_missing_params);
  2. runs the tower on a few seeded inputs through ImageBind(model_path) -- the reference's own constructor
     signature -- and reports embedding norms (vision: 1; audio: <= 20; text: exp(log_logit_scale)) and finiteness;
The comparison against the fp32 CPU oracle on the same weights lives with the other test infrastructure
(tests/checkpoint_vs_oracle.py calls validate() below and adds the cosine, tolerance >= 1 - 2e-4): nothing under
tools/ or the package imports oracle/.

Exit code 0 when every tower loads with no missing and no unused key and every embedding is finite.
Needs a GPU.  No network access is attempted.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def key_report(state_dict, tower: str, depth: int = 0):
    """Which checkpoint keys the tower takes, ignores and lacks -- computed from the library's own slot table."""
    import torch
    from hippomm_amd import _lib
    from hippomm_amd.encoder import HipTower, _PREFIXES
    mine = [k for k in state_dict if k.startswith(_PREFIXES[tower])]
    consumed, unused, errors = [], [], {}
    lib = _lib.load()
    import ctypes as C
    handle = C.c_void_p()
    _lib.check(lib.hmm_encoder_create(C.byref(handle), {"vision": 0, "audio": 1, "text": 2}[tower], depth), "create")
    try:
        for k in mine:
            if k.endswith(".mask"):                       # text preprocessor's causal-mask buffer: built into the kernel
                consumed.append(k)
                continue
            if depth > 0 and ".blocks." in k and int(k.split(".blocks.")[1].split(".")[0]) >= depth:
                continue                                  # --depth: blocks beyond the cut are not part of the test tower
            t = state_dict[k].detach().to("cuda", torch.float32).contiguous()
            rc = lib.hmm_encoder_load_param(handle, k.encode(), t.data_ptr(), t.numel(), _lib.stream_ptr())
            if rc == 0:
                consumed.append(k)
            else:
                unused.append(k)
                errors[k] = lib.hmm_last_error().decode()
        torch.cuda.synchronize()
        n_missing = lib.hmm_encoder_missing_params(handle)
        missing_msg = lib.hmm_last_error().decode() if n_missing else ""
    finally:
        lib.hmm_encoder_destroy(handle)
    return {"keys_under_prefix": len(mine), "consumed": len(consumed), "unused": unused, "unused_reasons": errors,
            "missing_count": int(n_missing), "missing": missing_msg}


def seeded_inputs(batch: int):
    import torch
    g = torch.Generator().manual_seed(0)
    return {"vision": torch.randn(batch, 3, 224, 224, generator=g),
            "audio": torch.randn(batch, 3, 1, 128, 204, generator=g),
            "text": torch.tensor([[49406, 320, 1125, 49407] + [0] * 73] * batch)}


def validate(checkpoint: str, towers, batch: int = 2, depth: int = 0):
    """-> (report, ok, state_dict, inputs, embeddings of the towers that loaded cleanly)."""
    import torch
    from hippomm_amd.encoder import ImageBind
    sd = torch.load(checkpoint, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    report, ok = {"checkpoint": checkpoint, "n_keys": len(sd)}, True
    other = sorted({k.split(".")[1] for k in sd if k.count(".") >= 2} - set(towers))
    report["modalities_in_file_not_built"] = other            # depth / thermal / imu: not on the reference's path
    for tower in towers:
        report[tower] = key_report(sd, tower, depth)
        ok &= report[tower]["missing_count"] == 0 and not report[tower]["unused"]
    inputs = seeded_inputs(batch)
    loadable = [t for t in towers if report[t]["missing_count"] == 0 and not report[t]["unused"]]
    emb = {}
    if loadable:
        cut = {t: depth for t in loadable} if depth > 0 else None
        model = ImageBind(checkpoint, towers=tuple(loadable), depth=cut)      # the reference's constructor argument
        out = model.forward({t: inputs[t].cuda() for t in loadable})
        for t in loadable:
            emb[t] = out[t].float().cpu()
            report[t]["embedding_norms"] = [round(float(x), 5) for x in emb[t].norm(dim=1)]
            report[t]["finite"] = bool(torch.isfinite(emb[t]).all())
            ok &= report[t]["finite"]
    return report, bool(ok), sd, inputs, emb


def parse_args(extra=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("--towers", nargs="+", default=["vision", "audio", "text"], choices=["vision", "audio", "text"])
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--depth", type=int, default=0, help="testing only: cut every tower to this many blocks")
    return ap.parse_args(extra)


def main():
    args = parse_args()
    report, ok, _, _, _ = validate(args.checkpoint, args.towers, args.batch, args.depth)
    report["ok"] = ok
    print(json.dumps(report, indent=1))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
