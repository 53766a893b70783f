"""The HIP path (through the C ABI) against the committed dense golden fixtures -- no oracle, no gcc on the box:
tests/golden/dense_fixtures.npz holds the CPU oracle's outputs for every BASELINE.json config at reduced batch
(tests/golden/make_dense_fixtures.py; the inputs are regenerated from tests/problems.py's seeded streams).
Tolerances are the ones DESIGN.md section 2 states per plan.  Needs an MI355X."""
import os

import numpy as np
import pytest

import altro_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from tests import golden_cases   # problem definitions only: nothing there imports the oracle
    return golden_cases


@pytest.fixture(scope="module")
def dense():
    return np.load(os.path.join(ROOT, "tests", "golden", "dense_fixtures.npz"))


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("name,dtype,flags,tol", [
    ("c1", altro_amd.F64, 0, 1e-9),                       # plan MFMA16 fp64: K, d <= 1e-8 absolute (north star), all 1e-9 relative
    ("c4", altro_amd.F32, 0, 2e-5),                       # fp32 storage, fp64 tiles
    ("c4", altro_amd.F32, altro_amd.F32_PURE, 5e-4),      # pure fp32
    ("c2shape", altro_amd.F64, 0, 0.0),                   # plan LANE: bit-identical
    ("c3shape", altro_amd.F64, 0, 0.0),
])
def test_tvlqr_against_dense_fixture(mk, dense, name, dtype, flags, tol):
    pr = mk.tvlqr_problem(name)
    assert mk.checksum(pr) == dense["tvlqr_%s_input_checksum" % name]
    batch = pr["A"].shape[0]
    bt = altro_amd.Batch(pr["N"], pr["n"], pr["m"], batch, dtype=dtype, flags=flags)
    bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
    bt.set_initial_state(pr["x0"]); bt.sweep()
    assert (bt.get("status") == -1).all()
    knots = mk.P_KNOTS.get(name)
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        got = bt.get(k)
        if k == "P" and knots is not None:
            got = got[:, knots]
        ref = dense["tvlqr_%s_%s" % (name, k)]
        if tol == 0.0:
            assert np.array_equal(got, ref), (name, k)
        else:
            assert relerr(got, ref) < tol, (name, k, relerr(got, ref))
    if name == "c1":   # the north star's own statement: gains within 1e-8 of the CPU path, absolute
        assert np.abs(bt.get("K") - dense["tvlqr_c1_K"]).max() < 1e-8 and np.abs(bt.get("d") - dense["tvlqr_c1_d"]).max() < 1e-8
    dv_tol = {0.0: 1e-15, 1e-9: 1e-9, 2e-5: 1e-4, 5e-4: 2e-3}[tol]
    assert relerr(bt.get("delta_V"), dense["tvlqr_%s_dV" % name]) <= dv_tol
    bt.close()


@pytest.mark.parametrize("name,tol", [("di_n10", 1e-11), ("di_n50", 1e-11), ("pendulum", 2e-7), ("bicycle", 5e-5)])
def test_solves_against_dense_fixture(mk, dense, name, tol):
    """Whole device solves (plan LANE, device models) land on the oracle's trajectories with the oracle's iteration
    counts: configs[0] at N = 10 and N = 50 (double_integrator_test.cpp:69-85), the pendulum and the bicycle."""
    c = mk.solve_case(name)
    x0s = np.asarray(c["x0s"], dtype=float)
    bt = altro_amd.Batch(c["N"], c["n"], c["m"], x0s.shape[0])
    assert bt.plan == altro_amd.PLAN_LANE
    bt.set_model(c["model"], c["h"])
    bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]).astype(float), np.asarray(c["Rd"], dtype=float)[None],
                         np.stack([c["xf"], c["xf"]]).astype(float), np.zeros((1, c["m"])), k_stride_zero=True,
                         batch_stride_zero=True)
    bt.set_initial_state(x0s)
    bt.set_input_guess(np.asarray(c["u0"], dtype=float)[None, None], k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=c["itmax"], use_backtracking=bool(c["bt"]))
    assert res["status"].tolist() == dense["solve_%s_status" % name].tolist()
    assert res["iterations"].tolist() == dense["solve_%s_iterations" % name].tolist()
    x, u = bt.get_nominal()
    assert relerr(x, dense["solve_%s_x" % name]) < tol
    assert relerr(u, dense["solve_%s_u" % name]) < tol * 10
    bt.close()
