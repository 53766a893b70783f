"""Plan LANE's backward sweep with sixteen (n <= 2: eight) lanes per problem and the records staged through LDS
(kernels/tvlqr_hex.hip) against the sweeps it replaces (ALTRO_HIP_LANE_HEX=0: four lanes per problem for (4, 2) / (2, 1), one lane
otherwise) and against the oracle.  Every output element of tvlqr.cpp:125-191 is still its own index-ordered dot product without
contraction, the factorisation keeps the IEEE division and square root: nothing may differ, bit for bit -- gains, cost-to-go,
Delta V, the failing knot point of an indefinite Quu, regularised or not -- for every shape with n <= 4, any batch (ragged last
waves: 4 or 8 problems per wave), any horizon (incl. shorter than the two-stage record ring)."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems
from tests.test_gpu_parity import run_hip, run_oracle, run_oracle_each

pytestmark = pytest.mark.gpu
KEYS = ("K", "d", "P", "p", "delta_V", "status")


def _env(name, value, fn):
    saved = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        os.environ.pop(name, None)
        if saved is not None:
            os.environ[name] = saved


SHAPES = [(n, m) for m in (1, 2, 3) for n in (1, 2, 3, 4)]


@pytest.mark.parametrize("n,m", SHAPES)
@pytest.mark.parametrize("batch,N", [(1, 1), (3, 2), (5, 3), (16, 5), (17, 8), (203, 50), (130, 101)])
@pytest.mark.parametrize("reg", [0.0, 0.37])
def test_hex_equals_lane_equals_oracle(batch, N, reg, n, m):
    pr = problems.random_ltv(batch, N, n, m, first=batch + N)
    hexr = _env("ALTRO_HIP_LANE_HEX", "1", lambda: run_hip(pr, altro_amd.PLAN_LANE, reg=reg))
    lane = _env("ALTRO_HIP_LANE_HEX", "0", lambda: run_hip(pr, altro_amd.PLAN_LANE, reg=reg))
    ref = run_oracle(pr, reg=reg)
    assert (hexr["status"] == -1).all()
    for k in KEYS + ("x", "u", "y"):
        assert np.array_equal(hexr[k], lane[k]), k
    for k in ("K", "d", "P", "p", "x", "u", "y"):
        assert np.array_equal(hexr[k], ref[k]), k
    assert np.array_equal(hexr["delta_V"], ref["dV"])


def test_hex_profile_names_the_kernel_it_ran():
    pr = problems.random_ltv(64, 6, 4, 2)
    bt = _env("ALTRO_HIP_LANE_HEX", "1", lambda: run_hip(pr, altro_amd.PLAN_LANE)["bt"])
    _env("ALTRO_HIP_LANE_HEX", "1", lambda: (bt.profile(1), bt.sweep(), bt.synchronize()))
    assert bt.profile_get(0)[2] == "hex_backward_kernel"
    assert bt.profile_get(1)[2] == "quad_forward_kernel"


@pytest.mark.parametrize("n,m", [(4, 2), (2, 1), (3, 3), (4, 1)])
def test_hex_failing_problems_stop_where_the_reference_stops(n, m):
    pr = problems.random_ltv(70, 10, n, m)
    for b, k in ((13, 4), (14, 9), (15, 0), (69, 7)):      # (three neighbours of one wave, and one in the ragged last wave)
        pr["R"][b, k] = -50.0 * np.eye(m).flatten()
    hexr = _env("ALTRO_HIP_LANE_HEX", "1", lambda: run_hip(pr, altro_amd.PLAN_LANE))
    lane = _env("ALTRO_HIP_LANE_HEX", "0", lambda: run_hip(pr, altro_amd.PLAN_LANE))
    ref = run_oracle_each(pr)
    assert hexr["status"].tolist() == ref["status"].tolist() == lane["status"].tolist()
    assert hexr["status"][13] == 4 and hexr["status"][14] == 9 and hexr["status"][15] == 0 and hexr["status"][69] == 7
    ok = hexr["status"] == -1
    for k in KEYS:     # (what a failed problem holds BELOW its failing knot point is whatever the buffer held: not compared)
        assert np.array_equal(hexr[k][ok], lane[k][ok]), k
    for b, kf in ((13, 4), (14, 9), (15, 0), (69, 7)):
        for k in ("K", "d"):
            assert np.array_equal(hexr[k][b, kf:], lane[k][b, kf:]) and np.array_equal(hexr[k][b, kf:], ref[k][b, kf:]), (k, b)
        assert np.array_equal(hexr["P"][b, kf + 1:], lane["P"][b, kf + 1:]) and np.array_equal(hexr["delta_V"][b], lane["delta_V"][b])
    assert np.array_equal(hexr["K"][ok], ref["K"][ok]) and np.array_equal(hexr["P"][ok], ref["P"][ok])
    assert np.array_equal(hexr["delta_V"][13], ref["dV"][13])


def test_hex_inside_the_batched_solve():
    """bicycle + steering bound (the C3 problem) and the pendulum on the launch-sequenced loop: whole solves bit-identical with either sweep"""
    from tests.test_gpu_fused import _bicycle, _pendulum, _same
    from tests.test_gpu_merit_split import _solve
    for make, opts in ((_bicycle(300), dict(iterations_max=40, use_backtracking=True)), (_pendulum(333), dict(iterations_max=30))):
        hexr = _env("ALTRO_HIP_FUSED", "0", lambda: _env("ALTRO_HIP_LANE_HEX", "1", lambda: _solve(make, {}, **opts)))
        lane = _env("ALTRO_HIP_FUSED", "0", lambda: _env("ALTRO_HIP_LANE_HEX", "0", lambda: _solve(make, {}, **opts)))
        _same(hexr, lane)
