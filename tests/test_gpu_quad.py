"""Plan LANE, (4, 2) and (2, 1): the backward sweep with four lanes per problem (kernels/tvlqr_quad_body.inc: rows /
columns dealt to the lanes; tvlqr_quad2_body.inc: one lane per element of the 2 x 2 objects) against the
lane-per-problem sweep it replaces (ALTRO_HIP_LANE_QUAD=0) and against the oracle.  The elements of every product of
tvlqr.cpp:125-191 are dealt to the four lanes, each still its own index-ordered dot product: nothing may differ, bit for
bit -- gains, cost-to-go, Delta V, the failing knot point of an indefinite Quu, regularised or not, f64 and f32."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems
from tests.test_gpu_parity import run_hip, run_oracle, run_oracle_each

pytestmark = pytest.mark.gpu
KEYS = ("K", "d", "P", "p", "delta_V", "status")


def _lane(fn):
    saved = os.environ.get("ALTRO_HIP_LANE_QUAD")
    os.environ["ALTRO_HIP_LANE_QUAD"] = "0"
    try:
        return fn()
    finally:
        os.environ.pop("ALTRO_HIP_LANE_QUAD", None)
        if saved is not None:
            os.environ["ALTRO_HIP_LANE_QUAD"] = saved


@pytest.mark.parametrize("n,m", [(4, 2), (2, 1)])
@pytest.mark.parametrize("batch,N", [(1, 1), (3, 2), (16, 5), (17, 8), (200, 50), (1000, 33), (130, 101)])
@pytest.mark.parametrize("reg", [0.0, 0.37])
def test_quad_equals_lane_equals_oracle(batch, N, reg, n, m):
    pr = problems.random_ltv(batch, N, n, m, first=batch + N)
    quad = run_hip(pr, altro_amd.PLAN_LANE, reg=reg)
    lane = _lane(lambda: run_hip(pr, altro_amd.PLAN_LANE, reg=reg))
    ref = run_oracle(pr, reg=reg)
    assert (quad["status"] == -1).all()
    for k in KEYS + ("x", "u", "y"):
        assert np.array_equal(quad[k], lane[k]), k
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        assert np.array_equal(quad[k], ref[k]), k
    assert np.array_equal(quad["delta_V"], ref["dV"])


@pytest.mark.parametrize("n,m,name", [(4, 2, "quad_backward_kernel"), (2, 1, "quad2_backward_kernel")])
def test_quad_profile_names_the_kernel_it_ran(n, m, name):
    pr = problems.random_ltv(64, 6, n, m)
    bt = run_hip(pr, altro_amd.PLAN_LANE)["bt"]
    bt.profile(1); bt.backward(); bt.synchronize()
    assert bt.profile_get(0)[2] == name
    assert bt.profile_get(1)[2] == ("quad_forward_kernel" if n == 4 else "lane_forward_kernel")   # (4, 2): the forward sweep too
    bt2 = _lane(lambda: run_hip(pr, altro_amd.PLAN_LANE)["bt"])
    assert bt2.profile_get(0)[2] == "lane_backward_kernel" and bt2.profile_get(1)[2] == "lane_forward_kernel"


@pytest.mark.parametrize("n,m", [(4, 2), (2, 1)])
def test_quad_failing_problems_stop_where_the_reference_stops(n, m):
    pr = problems.random_ltv(70, 10, n, m)
    for b, k in ((13, 4), (14, 9), (15, 0), (69, 7)):      # (13, 14, 15: three different quads' worth of one wave)
        pr["R"][b, k] = -50.0 * np.eye(m).flatten()
    quad = run_hip(pr, altro_amd.PLAN_LANE)
    lane = _lane(lambda: run_hip(pr, altro_amd.PLAN_LANE))
    ref = run_oracle_each(pr)
    assert quad["status"].tolist() == ref["status"].tolist() == lane["status"].tolist()
    assert quad["status"][13] == 4 and quad["status"][14] == 9 and quad["status"][15] == 0 and quad["status"][69] == 7
    ok = quad["status"] == -1
    for k in KEYS:     # (what a failed problem holds BELOW its failing knot point is whatever the buffer held: not compared)
        assert np.array_equal(quad[k][ok], lane[k][ok]), k
    for b, kf in ((13, 4), (14, 9), (15, 0), (69, 7)):
        for k in ("K", "d"):
            assert np.array_equal(quad[k][b, kf:], lane[k][b, kf:]) and np.array_equal(quad[k][b, kf:], ref[k][b, kf:]), (k, b)
        assert np.array_equal(quad["P"][b, kf + 1:], lane["P"][b, kf + 1:]) and np.array_equal(quad["delta_V"][b], lane["delta_V"][b])
    assert np.array_equal(quad["K"][ok], ref["K"][ok]) and np.array_equal(quad["P"][ok], ref["P"][ok])
    assert np.array_equal(quad["K"][13, 4:], ref["K"][13, 4:]) and np.array_equal(quad["d"][13, 4:], ref["d"][13, 4:])
    assert np.array_equal(quad["delta_V"][13], ref["dV"][13])


@pytest.mark.parametrize("n,m", [(4, 2), (2, 1)])
def test_quad_f32_storage(n, m):
    pr = problems.random_ltv(100, 20, n, m)

    def run():
        bt = altro_amd.Batch(20, n, m, 100, dtype=altro_amd.F32)
        bt.set_dynamics(pr["A"], pr["B"], pr["f"]); bt.set_cost(pr["Q"], pr["R"], pr["H"], pr["q"], pr["r"])
        bt.set_initial_state(pr["x0"]); bt.sweep()
        return {k: bt.get(k) for k in KEYS + ("x", "u")}
    quad, lane = run(), _lane(run)
    for k in quad:
        assert np.array_equal(quad[k], lane[k]), k


def test_quad_inside_the_batched_solve():
    """bicycle + steering bound, the C3 problem: the whole AL-iLQR solve is bit-identical with either sweep"""
    from tests.test_gpu_fused import _bicycle, _same
    from tests.test_gpu_merit_split import _solve
    from tests.test_gpu_fused import _pendulum
    for make, opts in ((_bicycle(300), dict(iterations_max=40, use_backtracking=True)), (_pendulum(333), dict(iterations_max=30))):
        quad = _solve(make, {}, **opts)
        lane = _lane(lambda: _solve(make, {}, **opts))
        _same(quad, lane)
