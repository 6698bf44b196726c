"""CPU: the oracle restatements reproduce the outputs the unmodified reference
produced for the committed recipes (tests/golden/make_golden.py)."""
import json
from pathlib import Path

import numpy as np
import pytest

import recipes
from oracle.consolidation_oracle import (evaluated_margin, select_key_frames_exact,
                                         select_key_frames_oracle)
from oracle.vector_ops_oracle import (top_k_cosine_similarity_oracle, top_k_documented_order)

GOLD = Path(__file__).resolve().parent / "golden"
SCAN = json.loads((GOLD / "scan_golden.json").read_text())["cases"]
SELECT = json.loads((GOLD / "select_golden.json").read_text())["cases"]
SELECT_INBAND = json.loads((GOLD / "select_golden.json").read_text())["inband_cases"]


def _sims(case):
    return np.array([np.nan if s is None else s for s in case["sims"]], dtype=np.float64)


@pytest.mark.parametrize("name", recipes.SCAN_CASES)
def test_scan_oracle_matches_reference(name):
    case = SCAN[name]
    q, store, k = recipes.scan_case(name)
    assert recipes.sha256(q, store) == case["input_sha256"], "recipe drifted from the golden inputs"
    with np.errstate(invalid="ignore", divide="ignore"):
        idx, sims = top_k_cosine_similarity_oracle(q, store, k)
    assert idx.dtype == np.int64
    assert str(sims.dtype) == case["sims_dtype"]
    assert idx.tolist() == case["indices"]
    np.testing.assert_allclose(sims, _sims(case), rtol=0, atol=2e-7, equal_nan=True)


@pytest.mark.parametrize("name", recipes.SCAN_CASES)
def test_documented_order_agrees_with_reference(name):
    """The tie / NaN rule the HIP path implements reproduces the reference's observed order."""
    case = SCAN[name]
    q, store, k = recipes.scan_case(name)
    store2 = store.reshape(1, -1) if store.ndim == 1 else store
    with np.errstate(invalid="ignore", divide="ignore"):
        sims = (store2 @ q) / (np.linalg.norm(store2, axis=1) * np.linalg.norm(q))
    idx, _ = top_k_documented_order(sims.astype(np.float32), k)
    assert idx.tolist() == case["indices"]


@pytest.mark.parametrize("name", recipes.SELECT_CASES)
def test_select_oracle_matches_reference(name):
    case = SELECT[name]
    f, t = recipes.select_case(name)
    assert recipes.sha256(f) == case["input_sha256"], "recipe drifted from the golden inputs"
    with np.errstate(invalid="ignore", divide="ignore"):
        kept = select_key_frames_oracle(f, t)
    assert kept.dtype == np.int64
    assert kept.tolist() == case["kept"]


@pytest.mark.parametrize("name", recipes.SELECT_CASES)
def test_select_exact_definition_matches_reference(name):
    """fp64-accumulated definition (what the HIP kernels compute) == reference on every fixture."""
    case = SELECT[name]
    f, _ = recipes.select_case(name)
    kept = select_key_frames_exact(f)
    assert kept.tolist() == case["kept"]
    if case["min_evaluated_margin"] is not None:
        with np.errstate(invalid="ignore", divide="ignore"):
            assert abs(evaluated_margin(f) - case["min_evaluated_margin"]) < 1e-6
        assert case["min_evaluated_margin"] > 5e-6, "fixture sits inside the BLAS-order band"


@pytest.mark.parametrize("name", recipes.SELECT_INBAND_CASES)
def test_select_inband_fixtures(name):
    """Pairs at float32(0.9) +- 1e-6 ... +- 3e-8.  What is pinned is the fp64-accumulated definition (the one
    the HIP path implements); the reference's own answer inside the band is BLAS-order dependent and stored as
    information only.  At +- 1e-6 (outside the band) definition and reference agree."""
    case = SELECT_INBAND[name]
    f, t = recipes.select_case(name)
    assert recipes.sha256(f) == case["input_sha256"], "recipe drifted from the golden inputs"
    assert select_key_frames_exact(f).tolist() == case["kept_exact_definition"]
    if name == "n24_inband_1e6":
        assert case["kept_reference_on_this_host"] == case["kept_exact_definition"]
        assert select_key_frames_oracle(f, t).tolist() == case["kept_reference_on_this_host"]
        assert max(abs(abs(x) - y) for x, y in zip(case["pair_minus_threshold_fp64"], [1e-6, 1e-6, 2e-6, 2e-6])) < 1e-7
    else:
        assert min(abs(x) for x in case["pair_minus_threshold_fp64"]) < 3e-8      # really inside the band


def test_select_threshold_is_float32():
    """0.9 is compared as float32(0.9) (hippocampal_memory.py:960 on a float32 gram)."""
    assert float(np.float32(0.9)) < 0.9          # 0.89999998
    s = np.array([np.float32(0.9)], dtype=np.float32)
    assert not np.all(s < 0.9)


# ---------------------------------------------------------------- encoder oracle regression fixture
def test_encoder_oracle_regression_fixture():
    """tests/golden/encoder_golden.json: the encoder oracle's own outputs on seeded weights / inputs (2-block towers).
    Guards the unpinned oracle against drift (recipe, code or torch changes); it says nothing about the reference."""
    import json
    from pathlib import Path
    import torch
    sys_path = str(Path(__file__).resolve().parent / "golden")
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("make_encoder_golden", sys_path + "/make_encoder_golden.py")
    mk = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mk)
    gold = json.loads((Path(sys_path) / "encoder_golden.json").read_text())
    for name, spec, seed, x in mk.cases():
        st, y = mk.run(name, spec, seed, x)
        case = gold[name]
        # weights_sha256 is informative only: trunc-normal sampling differs in the last bit between host CPUs
        want = torch.tensor(case["embeddings"], dtype=torch.float32)
        assert y.shape == want.shape
        assert (y - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), name
