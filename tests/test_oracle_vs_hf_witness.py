"""CPU: the committed full-geometry witness (tests/golden/hf_witness.json: HuggingFace CLIP models carrying the oracle's seeded
weights, 32 vision / 24 text layers) decodes, names the weights and inputs this environment regenerates from the seeds, and the
fp32 oracle text tower reproduces HF's text embeddings at all 24 blocks.  (The 32-block vision comparison takes a minute of CPU
and is recorded in the fixture by the generating script: max |diff| 1e-7; the GPU suite compares the HIP towers with both.)"""
import base64
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

from oracle import imagebind_oracle as ib

GOLDEN = Path(__file__).resolve().parent / "golden"


def test_witness_fixture_and_the_text_oracle_at_full_depth():
    sys.path.insert(0, str(GOLDEN))
    import make_hf_witness as mk
    fx = json.loads((GOLDEN / "hf_witness.json").read_text())
    for name, n_rows in (("vision", 2), ("text", 3)):
        rec = fx[name]
        emb = np.frombuffer(base64.b64decode(rec["hf_embeddings_b64"]), dtype="<f4").reshape(rec["shape"])
        assert emb.shape == (n_rows, 1024) and np.isfinite(emb).all()
        assert rec["oracle_vs_hf_max_abs_diff"] <= 2e-5 and rec["oracle_vs_hf_min_cos"] >= 1 - 1e-6
    assert hashlib.sha256(mk.vision_inputs().numpy().tobytes()).hexdigest() == fx["vision"]["input_sha256"]
    ids = mk.text_inputs()
    assert hashlib.sha256(ids.numpy().tobytes()).hexdigest() == fx["text"]["input_sha256"]
    rec = fx["text"]
    st = ib.synthetic_state(ib.TEXT_HUGE, seed=rec["weight_seed"], init=fx["init"])
    mk.check_weight_probe(st, rec)          # the SHA holds on the generating host only (trunc_normal_ is not bit-portable across CPUs)
    want = torch.from_numpy(np.frombuffer(base64.b64decode(rec["hf_embeddings_b64"]), dtype="<f4").reshape(rec["shape"]).copy())
    got = ib.text_forward(ids, st)
    assert (got - want).abs().max().item() <= 3e-4 and torch.nn.functional.cosine_similarity(got, want).min().item() >= 1 - 1e-6
    np.testing.assert_allclose(want.norm(dim=1).numpy(), rec["logit_scale"], rtol=1e-5)
