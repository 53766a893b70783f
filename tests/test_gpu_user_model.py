"""Caller-supplied dynamics in the batched loop (altro_hip_set_model_source): the device-side form of
ALTROSolver::SetExplicitDynamics (altro_solver.cpp:68-81) -- the caller's continuous dynamics and Jacobian as HIP source,
compiled by hiprtc into the library's own lane-per-problem kernels.  Pinned by supplying the reference's pendulum
(test/test_utils.cpp:43-82) as source: the solve is the compiled-in MODEL_PENDULUM's bit for bit; a model the library does
not ship (a unicycle) is checked against numpy.  Needs an MI355X."""
import os

import numpy as np
import pytest

import altro_amd
from tests import problems

pytestmark = pytest.mark.gpu

PENDULUM_SRC = r"""
// test/test_utils.cpp:43-82 (l = 0.5, g = 9.81, b = 0.1, m = 1): the expressions of the library's own pendulum_f / pendulum_J
template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xdot) {
  const T l = T(0.5), g = T(9.81), b = T(0.1), mm = T(1.0) * l * l;
  T s, c;
  if constexpr (sizeof(T) == 4) sincosf(x[0], &s, &c); else sincos(x[0], &s, &c);
  xdot[0] = x[1];
  xdot[1] = u[0] / mm - g * s / l - b * x[1] / mm;
}
template <typename T> __device__ void altro_user_jacobian(const T* x, const T* u, T* J) {
  (void)u;
  const T l = T(0.5), g = T(9.81), b = T(0.1), mm = T(1.0) * l * l;
  T s, c;
  if constexpr (sizeof(T) == 4) sincosf(x[0], &s, &c); else sincos(x[0], &s, &c);
  J[0] = T(0);
  J[1] = -g * c / l;
  J[2] = T(1);
  J[3] = -b / mm;
  J[4] = T(0);
  J[5] = T(1) / mm;
}
"""

UNICYCLE_SRC = r"""
// x = (px, py, theta), u = (v, omega):  px' = v cos(theta), py' = v sin(theta), theta' = omega
template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xdot) {
  xdot[0] = u[0] * cos(x[2]);
  xdot[1] = u[0] * sin(x[2]);
  xdot[2] = u[1];
}
template <typename T> __device__ void altro_user_jacobian(const T* x, const T* u, T* J) {   // 3 x 5, column-major
  for (int e = 0; e < 15; ++e) J[e] = T(0);
  J[0 + 2 * 3] = -u[0] * sin(x[2]);
  J[1 + 2 * 3] = u[0] * cos(x[2]);
  J[0 + 3 * 3] = cos(x[2]);
  J[1 + 3 * 3] = sin(x[2]);
  J[2 + 4 * 3] = T(1);
}
"""


def _pendulum(batch, dtype, source):
    N, n, m = 50, 2, 1
    h = np.float32(np.float32(3.0) / 50.0)
    bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
    if source:
        bt.set_model_source(PENDULUM_SRC, h)
    else:
        bt.set_model(altro_amd.MODEL_PENDULUM, h)
    xf = np.array([np.pi, 0.0])
    bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, m)),
                         k_stride_zero=True, batch_stride_zero=True)
    x0 = np.zeros((batch, n)); x0[:, 0] = problems.uniform01((batch,), 41) - 0.5
    bt.set_initial_state(x0)
    bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
    return bt


@pytest.mark.parametrize("dtype", [altro_amd.F64, altro_amd.F32])
def test_pendulum_as_source_equals_the_compiled_in_model(dtype):
    """Rollout, expansion (A, B, lx, lu), merit function and whole solves with a goal constraint: the same bits."""
    batch = 130
    res, xs, exps = [], [], []
    for source in (True, False):
        bt = _pendulum(batch, dtype, source)
        bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
        phi, dphi = bt.merit(np.linspace(0.0, 1.2, batch))
        exps.append((phi, dphi) + bt.get_expansion())
        G = np.zeros((2, 3)); G[0, 0] = 1.0; G[1, 1] = 1.0
        bt.add_linear_constraint(50, 50, altro_amd.CONE_EQUALITY, G, np.array([np.pi, 0.0]))
        bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
        os.environ["ALTRO_HIP_FUSED"] = "0"           # both on the launch-sequenced loop (the fused kernel has its own tests)
        try:
            res.append(bt.ilqr_solve(iterations_max=40))
        finally:
            del os.environ["ALTRO_HIP_FUSED"]
        xs.append(bt.get_nominal())
        bt.close()
    for a, b in zip(exps[0], exps[1]):
        assert np.array_equal(a, b)
    for k in ("status", "iterations", "phi", "stationarity", "feasibility", "alpha", "dual_updates"):
        assert np.array_equal(res[0][k], res[1][k]), k
    assert np.array_equal(xs[0][0], xs[1][0]) and np.array_equal(xs[0][1], xs[1][1])
    assert (res[0]["status"] == 0).sum() > batch // 2


def test_pendulum_as_source_against_the_fused_solve():
    """... and the policy's own path for MODEL_PENDULUM (the one-launch fused kernel) lands on the same bits too."""
    a = _pendulum(64, altro_amd.F64, True)
    b = _pendulum(64, altro_amd.F64, False)
    ra, rb = a.ilqr_solve(iterations_max=30), b.ilqr_solve(iterations_max=30)
    for k in ("status", "iterations", "phi", "stationarity"):
        assert np.array_equal(ra[k], rb[k]), k
    assert np.array_equal(a.get_nominal()[0], b.get_nominal()[0])
    a.close(); b.close()


def _unicycle_step(x, u, h):
    f = lambda x_, u_: np.array([u_[0] * np.cos(x_[2]), u_[0] * np.sin(x_[2]), u_[1]])
    xm = x + 0.5 * h * f(x, u)
    return x + h * f(xm, u)


def test_a_model_the_library_does_not_ship():
    """A unicycle (n = 3, m = 2): the rollout and the discrete Jacobians the kernels form from the caller's source against
    numpy (explicit midpoint, central differences), and a batch of parking manoeuvres solved to convergence."""
    N, n, m, batch = 40, 3, 2, 96
    h = np.float32(0.1)
    bt = altro_amd.Batch(N, n, m, batch)
    assert bt.plan == altro_amd.PLAN_LANE
    bt.set_model_source(UNICYCLE_SRC, h)
    xf = np.array([2.0, 1.0, 0.0])
    bt.set_tracking_cost(np.array([[1e-2] * 3, [50.0] * 3]), np.array([[1e-2, 1e-2]]), np.stack([xf, xf]), np.zeros((1, m)),
                         k_stride_zero=True, batch_stride_zero=True)
    x0 = (problems.uniform01((batch, n), 43) - 0.5) * np.array([1.0, 1.0, 0.6])
    bt.set_initial_state(x0)
    u0 = np.array([0.5, 0.1])
    bt.set_input_guess(u0[None, None], k_stride_zero=True, batch_stride_zero=True)
    bt.open_loop_rollout(); bt.accept(); bt.expand()
    x = bt.get("x")
    A, B, _, _ = bt.get_expansion()
    hd = float(h)
    for b in (0, 50, 95):
        xr = x0[b].copy()
        for k in range(N):
            assert np.abs(x[b, k] - xr).max() < 1e-12
            eps = 1e-6
            Afd = np.stack([(_unicycle_step(xr + eps * e, u0, hd) - _unicycle_step(xr - eps * e, u0, hd)) / (2 * eps) for e in np.eye(3)], 1)
            Bfd = np.stack([(_unicycle_step(xr, u0 + eps * e, hd) - _unicycle_step(xr, u0 - eps * e, hd)) / (2 * eps) for e in np.eye(2)], 1)
            assert np.abs(A[b, k].reshape(3, 3).T - Afd).max() < 1e-8       # column-major blocks
            assert np.abs(B[b, k].reshape(2, 3).T - Bfd).max() < 1e-8
            xr = _unicycle_step(xr, u0, hd)
    res = bt.ilqr_solve(iterations_max=100)
    assert (res["status"] == 0).sum() >= batch - 4, (res["status"] != 0).sum()
    xN = bt.get_knot(N, want_u=False)[0]
    ok = res["status"] == 0
    assert np.abs(xN[ok] - xf).max() < 0.1
    bt.close()


def test_a_source_that_does_not_compile_says_why():
    bt = altro_amd.Batch(10, 2, 1, 4)
    with pytest.raises(altro_amd.AltroHipError) as e:
        bt.set_model_source("template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xdot) { xdot[0] = y; }", 0.1)
    assert "error" in str(e.value) and "user_model" in str(e.value)
    big = altro_amd.Batch(10, 20, 4, 4, dtype=altro_amd.F32)
    with pytest.raises(altro_amd.AltroHipError):
        big.set_model_source(PENDULUM_SRC, 0.1)       # plans GENERIC / MFMA32 take models on fp64 handles (tests/test_gpu_generic_model.py)
    f32 = altro_amd.Batch(10, 12, 4, 4, dtype=altro_amd.F32)
    with pytest.raises(altro_amd.AltroHipError):
        f32.set_model_source(PENDULUM_SRC, 0.1)       # plan MFMA16 takes models on fp64 records (tests/test_gpu_tile_model.py)
    tile = altro_amd.Batch(10, 12, 4, 4)
    with pytest.raises(altro_amd.AltroHipError) as e2:
        tile.set_model_source("template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xdot) { xdot[0] = y; }", 0.1)
    assert "error" in str(e2.value) and "user_model" in str(e2.value)
    bt.close(); big.close(); f32.close(); tile.close()


GOAL_SRC = PENDULUM_SRC + r"""
// block 0: the goal pin c = x - (pi, 0) (pendulum_test.cpp:117-203 writes it as a callback pair too)
template <typename T> __device__ void altro_user_constraint(int id, const T* x, const T* u, T* c) {
  (void)id; (void)u;
  c[0] = x[0] - T(3.141592653589793);
  c[1] = x[1];
}
template <typename T> __device__ void altro_user_constraint_jacobian(int id, const T* x, const T* u, T* J) {   // 2 x 3
  (void)id; (void)x; (void)u;
  J[0] = T(1); J[1] = T(0); J[2] = T(0); J[3] = T(1); J[4] = T(0); J[5] = T(0);
}
"""

OBSTACLE_SRC = UNICYCLE_SRC + r"""
// block 0: stay outside a disc of radius 0.4 around (1.0, 0.45):  r^2 - |p - c|^2 <= 0   (nonlinear, INEQUALITY)
template <typename T> __device__ void altro_user_constraint(int id, const T* x, const T* u, T* c) {
  (void)id; (void)u;
  const T dx = x[0] - T(1.0), dy = x[1] - T(0.45);
  c[0] = T(0.16) - dx * dx - dy * dy;
}
template <typename T> __device__ void altro_user_constraint_jacobian(int id, const T* x, const T* u, T* J) {   // 1 x 5
  (void)id; (void)u;
  J[0] = -T(2) * (x[0] - T(1.0)); J[1] = -T(2) * (x[1] - T(0.45)); J[2] = T(0); J[3] = T(0); J[4] = T(0);
}
"""


def test_user_constraint_that_is_linear_matches_the_linear_block():
    """The pendulum's goal pin as a run-time compiled constraint against the same pin as a linear block: same decisions,
    same trajectory (the linear one goes through the bound-type fast path, the compiled one through the general rows)."""
    batch = 70
    out = []
    for user in (True, False):
        N, n, m = 50, 2, 1
        h = np.float32(np.float32(3.0) / 50.0)
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model_source(GOAL_SRC if user else PENDULUM_SRC, h)
        xf = np.array([np.pi, 0.0])
        bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, m)),
                             k_stride_zero=True, batch_stride_zero=True)
        x0 = np.zeros((batch, n)); x0[:, 0] = problems.uniform01((batch,), 41) - 0.5
        bt.set_initial_state(x0)
        bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
        if user:
            bt.add_user_constraint(N, N, altro_amd.CONE_EQUALITY, 2, 0)
        else:
            G = np.zeros((2, 3)); G[0, 0] = 1.0; G[1, 1] = 1.0
            bt.add_linear_constraint(N, N, altro_amd.CONE_EQUALITY, G, xf)
        res = bt.ilqr_solve(iterations_max=40)
        out.append((res, bt.get_nominal()))
        bt.close()
    (ra, (xa, ua)), (rb, (xb, ub)) = out
    assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["iterations"], rb["iterations"])
    assert np.array_equal(ra["dual_updates"], rb["dual_updates"])
    assert np.abs(xa - xb).max() < 1e-9 and np.abs(ua - ub).max() < 1e-8
    ok = ra["status"] == 0
    assert ok.sum() > batch // 2 and (ra["feasibility"][ok] < 1e-4).all()


def test_nonlinear_user_constraint_obstacle():
    """A nonlinear inequality (a disc to stay out of) on the unicycle: the batch converges, every converged trajectory
    clears the disc to the feasibility tolerance, and the unconstrained solve of the same problems does cut through it."""
    N, n, m, batch = 40, 3, 2, 64
    h = np.float32(0.1)
    xf = np.array([2.0, 1.0, 0.0])
    x0 = np.zeros((batch, n)); x0[:, 1] = (problems.uniform01((batch,), 47) - 0.5) * 0.2
    worst = []
    for constrained in (False, True):
        bt = altro_amd.Batch(N, n, m, batch)
        bt.set_model_source(OBSTACLE_SRC, h)
        bt.set_tracking_cost(np.array([[1e-2] * 3, [50.0] * 3]), np.array([[1e-2, 1e-2]]), np.stack([xf, xf]), np.zeros((1, m)),
                             k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(x0)
        bt.set_input_guess(np.array([[[0.5, 0.1]]]), k_stride_zero=True, batch_stride_zero=True)
        if constrained:
            bt.add_user_constraint(1, N, altro_amd.CONE_INEQUALITY, 1, 0)
        res = bt.ilqr_solve(iterations_max=150, penalty_initial=10.0)
        x, _ = bt.get_nominal()
        clear = np.sqrt((x[:, :, 0] - 1.0) ** 2 + (x[:, :, 1] - 0.45) ** 2).min(axis=1)
        ok = res["status"] == 0
        assert ok.sum() >= batch - 6, (constrained, int(ok.sum()))
        worst.append(clear[ok].min())
        if constrained:
            assert (res["feasibility"][ok] < 1e-4).all()
            assert np.abs(x[ok][:, -1] - xf).max() < 0.15
        bt.close()
    assert worst[0] < 0.3 and worst[1] > 0.4 - 2e-3, worst
