#!/usr/bin/env python3
"""Run the UNMODIFIED reference on the seeded cases of live_cases.py and print its results as JSON (build container only:
needs /root/reference).  A subprocess of tests/test_oracle_vs_reference_live.py, so that the stub modules the reference's
imports need (make_golden.import_reference) never enter the test process."""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import live_cases  # noqa: E402
from make_golden import import_reference  # noqa: E402

topk_ref, select_ref = import_reference()
out = {"scan": [], "select": []}
for seed in range(live_cases.N_SCAN):
    q, store, k = live_cases.scan_case(seed)
    idx, sims = topk_ref(q, store, k)
    out["scan"].append({"idx": [int(i) for i in idx], "sims": [None if np.isnan(s) else float(s) for s in sims],
                        "dtype": str(np.asarray(sims).dtype)})
for seed in range(live_cases.N_SELECT):
    f, t, thr = live_cases.select_case(seed)
    kept = select_ref(None, f, t, thr)
    out["select"].append([int(i) for i in kept])
print(json.dumps(out))
