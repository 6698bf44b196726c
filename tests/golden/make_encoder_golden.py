"""Regression fixture for the encoder ORACLE (not a pin to the reference -- the oracle is unpinned, see its header):
seeded synthetic weights + seeded inputs -> the oracle's embeddings, committed so that a change of the oracle, of the
weight recipe or of the installed torch that moves the numbers is noticed.  Run in the build container:

    python tests/golden/make_encoder_golden.py
"""
import hashlib
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import imagebind_oracle as ib      # noqa: E402


def weights_sha(st):
    h = hashlib.sha256()
    for k in sorted(st):
        h.update(k.encode())
        h.update(st[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def cases():
    g = torch.Generator().manual_seed(0)
    yield "vision", ib.reduced(ib.VISION_HUGE, 2), 1234, torch.randn(2, 3, 224, 224, generator=g)
    yield "audio", ib.reduced(ib.AUDIO_HUGE, 2), 4321, torch.randn(1, 3, 1, 128, 204, generator=g)
    ids = torch.zeros(2, 77, dtype=torch.long)
    ids[0, :5] = torch.tensor([49406, 320, 1125, 539, 49407])
    ids[1, :9] = torch.tensor([49406, 1237, 533, 518, 2533, 1629, 530, 518, 49407])
    yield "text", ib.reduced(ib.TEXT_HUGE, 2), 77, ids


def run(name, spec, seed, x):
    st = ib.synthetic_state(spec, seed=seed, init="rich")
    fwd = {"vision": ib.vision_forward, "audio": ib.audio_forward, "text": ib.text_forward}[name]
    with torch.no_grad():
        y = fwd(x, st, spec)
    return st, y


if __name__ == "__main__":
    out = {"torch": torch.__version__}
    for name, spec, seed, x in cases():
        st, y = run(name, spec, seed, x)
        out[name] = {"depth": spec.depth, "weight_seed": seed, "init": "rich", "weights_sha256": weights_sha(st),
                     "input_sha256": hashlib.sha256(x.numpy().tobytes()).hexdigest(),
                     "embeddings": [[float(v) for v in row] for row in y]}
    path = Path(__file__).with_name("encoder_golden.json")
    path.write_text(json.dumps(out))
    print("wrote", path, path.stat().st_size, "bytes")
