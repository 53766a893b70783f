"""The dense golden fixtures (tests/golden/dense_fixtures.npz, made by tests/golden/make_dense_fixtures.py) against the
oracle as it is built today: the checker is regression-pinned, and the seeded generators the fixtures depend on have not
drifted.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_dense_fixtures as mk   # noqa: E402


@pytest.fixture(scope="module")
def dense():
    return np.load(os.path.join(ROOT, "tests", "golden", "dense_fixtures.npz"))


@pytest.mark.parametrize("name", mk.TVLQR)
def test_tvlqr_fixture_is_what_the_oracle_computes(dense, name):
    out = mk.oracle_tvlqr(name)
    assert out["input_checksum"] == dense["tvlqr_%s_input_checksum" % name], "tests/problems.py changed its stream"
    assert (out["status"] == -1).all()
    for k in ("K", "d", "P", "p", "x", "u", "y", "dV"):
        ref = dense["tvlqr_%s_%s" % (name, k)]
        assert np.abs(out[k] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (name, k)


@pytest.mark.parametrize("name", mk.SOLVES)
def test_solve_fixture_is_what_the_oracle_computes(dense, name):
    out = mk.oracle_solve(name)
    assert out["status"].tolist() == dense["solve_%s_status" % name].tolist()
    assert out["iterations"].tolist() == dense["solve_%s_iterations" % name].tolist()
    for k in ("x", "u"):
        ref = dense["solve_%s_%s" % (name, k)]
        assert np.abs(out[k] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (name, k)
    if name.startswith("di_"):   # configs[0]: Success within iterations_max = 3 (double_integrator_test.cpp:141-146)
        assert out["status"].tolist() == [0] and out["iterations"][0] <= 3
