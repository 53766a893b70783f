"""MeritFunction in three launches (kernels/ilqr_lane.hip: rollout | per-knot-point terms | sums and phi') against the
one-launch kernel it replaces on the launch-sequenced loop (ALTRO_HIP_MERIT_SPLIT=0): same expressions, same order of
summation, so NOTHING may differ -- phi, phi', candidate trajectories, expansions, whole solves, duals; f64 and f32
storage, with and without speculative trials."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems
from tests.test_gpu_fused import _bicycle, _di_cones, _pendulum, _same

pytestmark = pytest.mark.gpu

ENV = ("ALTRO_HIP_NO_FUSED", "ALTRO_HIP_FUSED", "ALTRO_HIP_FUSED_SWEEPS", "ALTRO_HIP_NO_SPECULATION", "ALTRO_HIP_MERIT_SPLIT")


def _with_env(env, fn):
    saved = {k: os.environ.get(k) for k in ENV}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _solve(make, env, **opts):
    def run():
        bt = make()
        res = bt.ilqr_solve(**opts)
        x, u = bt.get_nominal()
        return res, x, u, bt.get("x"), bt.get("K"), bt
    return _with_env(dict(env, ALTRO_HIP_FUSED="0"), run)


@pytest.mark.parametrize("speculation", [True, False])
@pytest.mark.parametrize("name,make,opts,duals", [
    ("pendulum, cubic line search", _pendulum(333), dict(iterations_max=30), []),
    ("bicycle + steering bound, backtracking", _bicycle(500), dict(iterations_max=40, use_backtracking=True),
     [(0, 0, 2), (20, 0, 2), (40, 0, 2)]),
    ("bicycle + steering bound, cubic", _bicycle(200), dict(iterations_max=40), [(7, 0, 2)]),
    ("double integrator, bounds", _di_cones("bounds", 70), dict(iterations_max=20, penalty_scaling=100.0), [(10, 0, 4), (3, 0, 4)]),
    ("double integrator, second-order cone", _di_cones("soc", 70), dict(iterations_max=30, penalty_scaling=100.0),
     [(10, 0, 4), (3, 0, 3)]),
])
def test_split_merit_solves_are_bit_identical(name, make, opts, duals, speculation):
    spec = {} if speculation else {"ALTRO_HIP_NO_SPECULATION": "1"}
    one = _solve(make, dict(spec, ALTRO_HIP_MERIT_SPLIT="0"), **opts)
    three = _solve(make, spec, **opts)
    _same(one, three)
    assert int(one[0]["merit_launches"]) == int(three[0]["merit_launches"]) > 0
    for (k, slot, p) in duals:
        assert np.array_equal(one[5].get_duals(k, slot, p), three[5].get_duals(k, slot, p))


@pytest.mark.parametrize("dtype", [altro_amd.F64, altro_amd.F32])
def test_split_merit_entry_point(dtype):
    """altro_hip_merit itself: phi, phi', the candidate trajectory (x_, y_, u_) and the refreshed expansion"""
    N, n, m, batch = 30, 4, 2, 130
    x_ref, u_ref = problems.bicycle_reference(N + 1)
    alpha = 0.25 + 0.5 * problems.uniform01((batch,), 77)

    def run():
        bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
        bt.set_model(altro_amd.MODEL_BICYCLE, np.float32(0.1))
        bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                             batch_stride_zero=True)
        G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
        bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, 0.2))
        bt.set_initial_state(x_ref[0] + (problems.uniform01((batch, n), 23, 0) - 0.5) * 0.6)
        bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
        bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
        out = [bt.merit(alpha, derivative=True)]
        out.append((bt.get("x"), bt.get("u"), bt.get("y")))
        out.append(bt.get_expansion())
        out.append(bt.merit(0.5, derivative=False)[:1])
        out.append((bt.get("x"), bt.get("u"), bt.get("y")))
        return out

    one = _with_env({"ALTRO_HIP_MERIT_SPLIT": "0"}, run)
    three = _with_env({}, run)
    for a, b in zip(one, three):
        for u, v in zip(a, b):
            assert np.array_equal(np.asarray(u), np.asarray(v))
