"""tools/validate_checkpoint.py on a synthetic .pth with upstream key names (there is no real imagebind_huge.pth here):
the tool loads it through ImageBind(model_path) and reports consumed / unused / missing keys per tower;
tests/checkpoint_vs_oracle.py adds the cosine against the fp32 oracle; a foreign key and a dropped key are reported
and fail the run."""
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from oracle import imagebind_oracle as ib

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(path, *extra, script=("tools", "validate_checkpoint.py")):
    r = subprocess.run([sys.executable, str(ROOT.joinpath(*script)), str(path), "--depth", "1", *extra],
                       capture_output=True, text=True, timeout=900)
    return r.returncode, json.loads(r.stdout[r.stdout.index("{"):])


def test_tool_on_a_synthetic_checkpoint(tmp_path):
    sd = {}
    for spec, seed in ((ib.VISION_HUGE, 1), (ib.AUDIO_HUGE, 2), (ib.TEXT_HUGE, 3)):
        sd.update(ib.synthetic_state(ib.reduced(spec, 1), seed=seed, init="rich"))
    sd["modality_preprocessors.text.mask"] = torch.zeros(77, 77)                     # upstream buffer: consumed silently
    sd["modality_trunks.depth.blocks.0.norm_1.weight"] = torch.ones(384)            # a modality that is not built
    good = tmp_path / "imagebind_huge.pth"
    torch.save(sd, good)
    rc, rep = _run(good, script=("tests", "checkpoint_vs_oracle.py"))
    assert rc == 0 and rep["ok"], rep
    assert rep["modalities_in_file_not_built"] == ["depth"]
    for t in ("vision", "audio", "text"):
        assert rep[t]["missing_count"] == 0 and rep[t]["unused"] == [] and rep[t]["finite"]
        assert min(rep[t]["cos_vs_fp32_oracle"]) >= 1 - 5e-5
    assert all(abs(n - 1.0) < 1e-4 for n in rep["vision"]["embedding_norms"])

    bad = dict(sd)
    del bad["modality_trunks.vision.blocks.0.mlp.fc2.bias"]
    bad["modality_trunks.audio.blocks.0.attn.layer_scale_gamma"] = torch.ones(768)  # a key the restated tower does not know
    p = tmp_path / "bad.pth"
    torch.save(bad, p)
    rc, rep = _run(p, "--towers", "vision", "audio")
    assert rc == 1 and not rep["ok"]
    assert rep["vision"]["missing_count"] == 1 and "fc2.bias" in rep["vision"]["missing"]
    assert rep["audio"]["unused"] == ["modality_trunks.audio.blocks.0.attn.layer_scale_gamma"]
