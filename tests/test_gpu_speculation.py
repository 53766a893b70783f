"""Speculative line-search trials (altro_hip_ilqr_solve, kernels/ilqr_types.h): the first step alpha0 = 1 rides in the
launch that evaluates phi(0), and the backtracking steps alpha beta^j are evaluated several per launch while the chip
has idle lanes.  The state machine consumes them in order, so NOTHING may change -- trajectories, iteration counts,
statuses, duals are compared bit for bit against the same solve with ALTRO_HIP_NO_SPECULATION=1 -- except the number
of merit launches."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu


def _solve_both(make, **opts):
    out = []
    os.environ["ALTRO_HIP_NO_FUSED"] = "1"    # speculation belongs to the launch-sequenced loop: test it there, alone
    for off in (True, False):
        if off:
            os.environ["ALTRO_HIP_NO_SPECULATION"] = "1"
        else:
            os.environ.pop("ALTRO_HIP_NO_SPECULATION", None)
        try:
            bt = make()
            res = bt.ilqr_solve(**opts)
            x, u = bt.get_nominal()
            out.append((res, x, u, bt))
        finally:
            os.environ.pop("ALTRO_HIP_NO_SPECULATION", None)
    os.environ.pop("ALTRO_HIP_NO_FUSED", None)
    return out


def _same(a, b):
    (ra, xa, ua, _), (rb, xb, ub, _) = a, b
    assert np.array_equal(xa, xb) and np.array_equal(ua, ub)
    for key in ("status", "iterations", "stationarity", "alpha", "phi", "feasibility", "penalty", "dual_updates"):
        if key in ra:
            assert np.array_equal(np.asarray(ra[key]), np.asarray(rb[key])), key
    assert int(ra["sweeps"]) == int(rb["sweeps"])


def _bicycle(batch, N=40):
    n, m, h = 4, 2, np.float32(0.1)
    x_ref, u_ref = problems.bicycle_reference(N + 1)

    def make():
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model(altro_amd.MODEL_BICYCLE, h)
        bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                             batch_stride_zero=True)
        G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
        bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
        bt.set_initial_state(x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * 0.6)
        bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
        return bt
    return make


@pytest.mark.parametrize("backtracking", [True, False])
def test_lane_speculation_changes_nothing_but_the_launch_count(backtracking):
    off, on = _solve_both(_bicycle(777), iterations_max=60, use_backtracking=backtracking)
    _same(off, on)
    assert int(on[0]["merit_launches"]) < int(off[0]["merit_launches"])
    for k in range(0, 41, 10):   # the duals went the same way too
        assert np.array_equal(off[3].get_duals(k, 0, 2), on[3].get_duals(k, 0, 2))


def _linear_12x4(batch, N=24):
    n, m, h = 12, 4, 0.05
    I4, Z4 = np.eye(4), np.zeros((4, 4))
    A = np.block([[I4, h * I4, 0.5 * h * h * I4], [Z4, I4, h * I4], [Z4, Z4, I4]])
    B = np.vstack([h ** 3 / 6 * I4, 0.5 * h * h * I4, h * I4])
    cm = lambda M: np.ascontiguousarray(M.T).reshape(1, 1, -1)   # noqa: E731
    Qd = np.concatenate([10.0 * np.ones(4), np.ones(4), 0.1 * np.ones(4)])
    x0 = np.concatenate([2.0 * problems.uniform01((batch, 4), 31, 0) - 1.0, np.zeros((batch, 8))], axis=1)

    def make():
        bt = altro_amd.Batch(N, n, m, batch)
        assert bt.plan == altro_amd.PLAN_MFMA16
        bt.set_dynamics(cm(A), cm(B), None, k_stride_zero=True, batch_stride_zero=True)
        bt.set_tracking_cost(np.tile(Qd, (1, N + 1, 1)), np.full((1, N, m), 1e-2), np.zeros((1, N + 1, n)), np.zeros((1, N, m)),
                             batch_stride_zero=True)
        G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
        bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2 * m, 3.0))
        bt.set_initial_state(x0)
        bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        return bt
    return make


@pytest.mark.parametrize("backtracking", [True, False])
def test_mfma16_speculation_changes_nothing_but_the_launch_count(backtracking):
    off, on = _solve_both(_linear_12x4(96), iterations_max=40, use_backtracking=backtracking)
    _same(off, on)
    assert int(on[0]["merit_launches"]) < int(off[0]["merit_launches"])
