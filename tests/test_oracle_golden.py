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


def test_round4_dense_cost_fixtures_are_what_the_oracle_computes(dense):
    """Row a9: the dense quadratic cost (ALTROSolver::SetQuadraticCost) in whole solves on both plans' shapes and in the
    merit / expansion rows, regenerated and compared."""
    for constrained in (False, True):
        tag = "al" if constrained else "lq"
        for name, fn in (("quad12", mk.oracle_quad12), ("quad4", mk.oracle_quad4)):
            out = fn(constrained)
            assert (out["status"] == 0).all()
            for k, v in out.items():
                ref = dense["%s_%s_%s" % (name, tag, k)]
                assert np.abs(np.asarray(v, dtype=float) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (name, constrained, k)
    for name in ("quad12", "quad4"):
        out = mk.oracle_merit(name)
        for k, v in out.items():
            ref = dense["merit_%s_%s" % (name, k)]
            assert np.abs(v - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (name, k)


@pytest.mark.parametrize("group", ["al", "mpc", "lq12", "merit", "reg"])
def test_round3_fixtures_are_what_the_oracle_computes(dense, group):
    """The constrained solves (3 / 5 / 9 iterations for the reference's own start), the MPC loop, the (12, 4) iLQR solves, the
    merit / expansion / stationarity rows and the regularised backward pass, regenerated and compared."""
    if group == "al":
        kats = mk.load_kats()
        for kind in mk.AL_KINDS:
            out = mk.oracle_al(kind)
            assert out["status"][0] == 0 and out["iterations"][0] == kats["double_integrator_constrained"][kind]["iterations"]
            for k, v in out.items():
                ref = dense["al_%s_%s" % (kind, k)]
                assert np.abs(np.asarray(v, dtype=float) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (kind, k)
    elif group == "mpc":
        out = mk.oracle_mpc()
        assert (out["status"] == 0).all()
        for k, v in out.items():
            ref = dense["mpc_%s" % k]
            assert np.abs(np.asarray(v, dtype=float) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k
    elif group == "lq12":
        for constrained in (False, True):
            out = mk.oracle_lq12(constrained)
            for k, v in out.items():
                ref = dense["lq12_%s_%s" % ("al" if constrained else "lq", k)]
                assert np.abs(np.asarray(v, dtype=float) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (constrained, k)
    elif group == "merit":
        for name in ("pendulum", "bicycle", "lq12"):
            out = mk.oracle_merit(name)
            for k, v in out.items():
                ref = dense["merit_%s_%s" % (name, k)]
                assert np.abs(v - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (name, k)
    else:
        out = mk.oracle_reg()
        assert (out["reg_status"] == -1).all() and out["bad_status"][1] >= 0 and out["bad_status"][0] == -1
        for k, v in out.items():
            ref = dense["tvlqr_%s" % k]
            assert np.abs(np.asarray(v, dtype=float) - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), k
