#!/usr/bin/env python3
"""One-off soak: plan MFMA16 (packed symmetric records, XCD-aware mapping) against the oracle on random shapes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import altro_amd
from tests import problems
from tests.test_gpu_parity import run_hip, run_oracle, relerr

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(40):
    N = int(rng.integers(1, 90)); batch = int(rng.integers(1, 140))
    pr = problems.random_ltv(batch, N, 12, 4, first=int(rng.integers(0, 1000)))
    if rng.random() < 0.3:
        pr["f"] = np.zeros_like(pr["f"])
    reg = float(rng.choice([0.0, 1e-3, 0.1]))
    dt = altro_amd.F64
    out = run_hip(pr, altro_amd.PLAN_MFMA16, reg=reg)
    ref = run_oracle(pr, reg=reg)
    assert (out["status"] == -1).all()
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        e = relerr(out[k], ref[k]); worst = max(worst, e)
        assert e < 1e-9, (it, N, batch, k, e)
print("ok, worst relative error %.2e" % worst)
