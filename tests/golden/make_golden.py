#!/usr/bin/env python3
"""Generate tests/golden/{scan,select}_golden.json by running the UNMODIFIED reference.

Run in the build container only (it needs /root/reference, which does not exist
on the GPU box):

    python tests/golden/make_golden.py

* ``top_k_cosine_similarity`` is imported directly from
  /root/reference/hippomm/utils/vector_ops.py (:151-188).
* ``HippocampalMemory._select_key_frames``
  (/root/reference/hippomm/core/hippocampal_memory.py:944-967) is imported behind
  empty stub modules registered for the third-party imports that are absent in
  this image (cv2, librosa, soundfile, openai, skimage, sklearn.cluster DBSCAN,
  token_count, imagebind, faster_whisper, decord, qwen_vl_utils, PIL, requests).
  The method never touches ``self``; it is called with ``self=None``.

Only inputs' recipes (tests/golden/recipes.py), input hashes and the reference's
outputs are written -- no reference source.
"""
from __future__ import annotations

import json
import platform
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
import recipes  # noqa: E402

REF = "/root/reference"


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules.setdefault(name, mod)
    return sys.modules[name]


def import_reference():
    sys.path.insert(0, REF)
    anything = type("Anything", (), {"__init__": lambda self, *a, **k: None})
    for name in ["cv2", "librosa", "soundfile", "decord", "requests", "token_count",
                 "faster_whisper", "qwen_vl_utils", "openai", "imagebind", "imagebind.models",
                 "imagebind.models.imagebind_model", "skimage", "skimage.metrics"]:
        _stub(name)
    sys.modules["imagebind"].data = types.SimpleNamespace()
    sys.modules["imagebind.models"].imagebind_model = sys.modules["imagebind.models.imagebind_model"]
    sys.modules["imagebind.models.imagebind_model"].ModalityType = type(
        "ModalityType", (), {"TEXT": "text", "VISION": "vision", "AUDIO": "audio"})
    sys.modules["openai"].OpenAI = anything
    sys.modules["faster_whisper"].WhisperModel = anything
    sys.modules["qwen_vl_utils"].process_vision_info = lambda *a, **k: None
    sys.modules["skimage.metrics"].structural_similarity = lambda *a, **k: 0.0
    sys.modules["token_count"].TokenCount = anything
    try:
        import PIL  # noqa: F401
    except Exception:
        _stub("PIL")
        _stub("PIL.Image", Image=anything)
        sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    try:
        import sklearn.cluster  # noqa: F401
    except Exception:
        _stub("sklearn")
        _stub("sklearn.cluster", DBSCAN=anything)
    from hippomm.utils.vector_ops import top_k_cosine_similarity
    from hippomm.core.hippocampal_memory import HippocampalMemory
    return top_k_cosine_similarity, HippocampalMemory._select_key_frames


def main():
    topk_ref, select_ref = import_reference()
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle.consolidation_oracle import evaluated_margin

    env = {"numpy": np.__version__, "python": platform.python_version(),
           "machine": platform.processor() or platform.machine()}

    scan = {"_generated_by": "tests/golden/make_golden.py", "_env": env,
            "_reference": "hippomm/utils/vector_ops.py:151-188", "cases": {}}
    for name in recipes.SCAN_CASES:
        q, store, k = recipes.scan_case(name)
        idx, sims = topk_ref(q, store, k)
        scan["cases"][name] = {
            "k": k, "store_shape": list(store.shape), "store_dtype": str(store.dtype),
            "input_sha256": recipes.sha256(q, store),
            "indices": [int(i) for i in idx],
            "sims": [None if np.isnan(s) else float(s) for s in sims],
            "sims_dtype": str(sims.dtype),
        }
    (HERE / "scan_golden.json").write_text(json.dumps(scan, indent=1))

    sel = {"_generated_by": "tests/golden/make_golden.py", "_env": env,
           "_reference": "hippomm/core/hippocampal_memory.py:944-967", "cases": {}}
    for name in recipes.SELECT_CASES:
        f, t = recipes.select_case(name)
        with np.errstate(invalid="ignore", divide="ignore"):
            kept = select_ref(None, f, t)
            margin = evaluated_margin(f)
        sel["cases"][name] = {
            "n": int(f.shape[0]), "threshold": 0.9,
            "input_sha256": recipes.sha256(f),
            "kept": [int(i) for i in kept], "kept_dtype": str(kept.dtype),
            "min_evaluated_margin": (None if not np.isfinite(margin) else margin),
        }
    # in-band fixtures: the reference's answer here depends on this host's BLAS summation order -> informational
    from oracle.consolidation_oracle import select_key_frames_exact, pair_similarities
    sel["inband_cases"] = {}
    for name in recipes.SELECT_INBAND_CASES:
        f, t = recipes.select_case(name)
        kept = select_ref(None, f, t)
        exact = select_key_frames_exact(f)
        n_pairs = 4 if name == "n24_inband_1e6" else 7
        sel["inband_cases"][name] = {
            "n": int(f.shape[0]), "threshold": 0.9, "input_sha256": recipes.sha256(f),
            "kept_reference_on_this_host": [int(i) for i in kept],
            "kept_exact_definition": [int(i) for i in exact],
            "informational": "BLAS-order dependent: the reference's kept list is what numpy %s on %s returned" % (
                np.__version__, env["machine"]),
            "pair_minus_threshold_fp64": [float(x) for x in pair_similarities(f, n_pairs) - float(np.float32(0.9))],
        }
    (HERE / "select_golden.json").write_text(json.dumps(sel, indent=1))
    # memory_store event JSON, exactly as save_theta_event writes it (hippocampal_memory.py:110-133, :331-335)
    from hippomm.core.hippocampal_memory import ThetaEvent
    ev = ThetaEvent(**recipes.event_case())
    (HERE / "event_golden.json").write_text(json.dumps(ev.to_dict(), indent=2))
    print("event golden:", (HERE / "event_golden.json").stat().st_size, "bytes")
    for name, c in sel["inband_cases"].items():
        print(f"inband {name:20s} ref={c['kept_reference_on_this_host'][:12]} exact={c['kept_exact_definition'][:12]} "
              f"pairs-thr={['%.1e' % x for x in c['pair_minus_threshold_fp64']]}")
    for name, c in sel["cases"].items():
        print(f"select {name:24s} n={c['n']:5d} kept={len(c['kept']):4d} margin={c['min_evaluated_margin']}")
    for name, c in scan["cases"].items():
        print(f"scan   {name:24s} idx={c['indices'][:6]} sims={c['sims'][:3]}")


if __name__ == "__main__":
    main()
