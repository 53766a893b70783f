"""Nonlinear dynamics on the device past the (12, 4) tile (missing #2): the compiled-in 13-state quaternion quadrotor
(csrc/models.h MODEL_QUADROTOR13; oracle/models_oracle.c pins its Jacobian by central differences) iterated by the iLQR loop of
plans GENERIC and MFMA32 -- ALTROSolver::SetExplicitDynamics-style dynamics (altro_solver.cpp:68-81, knotpoint_data.cpp:406-419,
:710-719) through the explicit midpoint rule of the reference's own test harness (test_utils.cpp:84-132) -- against the oracle's
restatement of SolverImpl with the same model.

* rollout, dynamics expansion A_k, B_k, cost expansion, gains of the backward sweep, one merit evaluation per problem (phi, phi', the
  candidate), the candidate's stationarity;
* whole (AL-)iLQR solves, free and with a thrust bound: status, iteration count, trajectories;
* a batched NMPC of 1024 vehicles: the first solve's statuses / iterations / trajectories against the oracle on a sample, then
  receding-horizon steps on the resident batch."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems

pytestmark = pytest.mark.gpu

N, n, m = 30, 13, 4
H = np.float32(0.02)
HOVER = np.array([0.5 * 9.81, 0.0, 0.0, 0.0])
PLANS = [altro_amd.PLAN_GENERIC, altro_amd.PLAN_MFMA32]


def make_case(batch, seed=0):
    """Fly from a perturbed state to hover at the origin with the identity attitude: tracking cost."""
    x0 = np.zeros((batch, n))
    x0[:, :3] = 0.8 * problems.normal((batch, 3), 191 + seed)
    q = np.concatenate([np.ones((batch, 1)), 0.12 * problems.normal((batch, 3), 192 + seed)], axis=1)
    x0[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    x0[:, 7:10] = 0.3 * problems.normal((batch, 3), 193 + seed)
    x0[:, 10:] = 0.2 * problems.normal((batch, 3), 194 + seed)
    xref = np.zeros(n); xref[3] = 1.0
    Qd = np.concatenate([np.full(3, 2.0), np.full(4, 1.0), np.full(3, 0.5), np.full(3, 0.1)])
    Rd = np.array([0.05, 20.0, 20.0, 20.0])
    return dict(x0=x0, Qd=Qd, Qfd=20.0 * Qd, Rd=Rd, xref=xref, uref=HOVER, u0=HOVER.copy())


def make_hip(c, plan):
    bt = altro_amd.Batch(N, n, m, c["x0"].shape[0], plan=plan)
    assert bt.plan == plan
    bt.set_model(altro_amd.MODEL_QUADROTOR13, H)
    bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]), c["Rd"][None], np.stack([c["xref"], c["xref"]]), c["uref"][None],
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(c["x0"])
    bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
    return bt


def make_oracle(c, b, blocks=()):
    s = oracle.ILQR(N, n, m, H, oracle.DYN_MODEL, oracle.MODEL_QUADROTOR13, cost_kind=oracle.COST_DIAGONAL)
    for k in range(N + 1):
        s.L.oracle_ilqr_set_lqr_cost(s.h, k, np.ascontiguousarray(c["Qfd"] if k == N else c["Qd"]), np.ascontiguousarray(c["Rd"]),
                                     np.ascontiguousarray(c["xref"]), np.ascontiguousarray(c["uref"]))
    s.L.oracle_ilqr_set_initial_state(s.h, np.ascontiguousarray(c["x0"][b]))
    for (k0, k1, cone, G, g) in blocks:
        for k in range(k0, k1 + 1):
            s.add_linear_constraint(k, cone, G, g)
    s.L.oracle_ilqr_initialize(s.h)
    for k in range(N):
        s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(c["u0"]))
    return s


@pytest.mark.parametrize("plan", PLANS)
def test_rollout_expansion_merit_stationarity(plan):
    batch = 9
    c = make_case(batch)
    bt = make_hip(c, plan)
    bt.open_loop_rollout(); bt.accept(); bt.expand()
    A0, B0, lx0, lu0 = bt.get_expansion()
    bt.backward()
    assert (bt.get("status") == -1).all()
    alphas = np.linspace(0.0, 1.0, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc, yc = bt.get("x"), bt.get("u"), bt.get("y")
    A1, B1, _, _ = bt.get_expansion()                 # the merit pass with derivative refreshed A, B at ITS candidate
    st = bt.stationarity()
    K = bt.get("K")
    for b in [0, 3, 8]:
        s = make_oracle(c, b)
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        np.testing.assert_allclose(A0[b], s.get("A"), rtol=1e-12, atol=1e-12)    # CalcDynamicsExpansion: A, B of the midpoint rule
        np.testing.assert_allclose(B0[b], s.get("B"), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(lx0[b], s.get("lx"), rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(lu0[b], s.get("lu"), rtol=1e-11, atol=1e-11)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        Kr = s.get("K")
        assert np.abs(K[b] - Kr).max() <= 1e-8 * max(1.0, np.abs(Kr).max())
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= 1e-10 * max(1.0, abs(p_ref)), (b, phi[b], p_ref)
        assert abs(dphi[b] - dp_ref) <= 1e-8 * max(1.0, abs(dp_ref)), (b, dphi[b], dp_ref)
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=2e-9, atol=2e-9)
        np.testing.assert_allclose(uc[b], s.get("u_cand"), rtol=2e-8, atol=2e-8)
        np.testing.assert_allclose(yc[b], s.get("y_cand"), rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(A1[b], s.get("A"), rtol=1e-8, atol=1e-8)      # (MeritFunction refreshed the oracle's too)
        np.testing.assert_allclose(B1[b], s.get("B"), rtol=1e-8, atol=1e-8)
        ref_st = s.L.oracle_ilqr_stationarity(s.h)
        assert abs(st[b] - ref_st) <= 1e-7 * max(1.0, ref_st), (b, st[b], ref_st)


@pytest.mark.parametrize("plan", PLANS)
@pytest.mark.parametrize("constrained,backtracking", [(False, False), (True, False), (False, True)])
def test_whole_solves(plan, constrained, backtracking):
    batch = 21
    c = make_case(batch)
    bt = make_hip(c, plan)
    blocks = []
    if constrained:
        Gb = np.zeros((2, n + m)); Gb[0, n] = 1.0; Gb[1, n] = -1.0
        blocks = [(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.array([1.25 * HOVER[0], -0.6 * HOVER[0]]))]
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
    res = bt.ilqr_solve(iterations_max=50, use_backtracking=backtracking, tol_stationarity=1e-3)
    x, u = bt.get_nominal()
    nconv = 0
    for b in [0, 10, 20]:
        s = make_oracle(c, b, blocks)
        if blocks:
            s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 50, 1e-3, 1e-4, 1e-8, 1 if backtracking else 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=2e-7, atol=2e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=2e-6, atol=2e-6)
    assert nconv >= 2


def test_batched_nmpc_of_1024_quaternion_quadrotors():
    """1024 vehicles on an ALTRO_HIP_PLAN_AUTO handle (plan MFMA32: the sweeps on 2 x 2 matrix-core tiles, the loop's model kernels
    of plan GENERIC): the first solve against the oracle on a seeded sample, then three receding-horizon steps on the resident batch
    (x_1 becomes the initial state, the trajectory is shifted, the solve is warm-started) whose first solve of the sample vehicles
    the oracle's own loop reproduces."""
    batch = 1024
    c = make_case(batch, seed=7)
    bt = altro_amd.Batch(N, n, m, batch)
    assert bt.plan == altro_amd.PLAN_MFMA32
    bt.set_model(altro_amd.MODEL_QUADROTOR13, H)
    bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]), c["Rd"][None], np.stack([c["xref"], c["xref"]]), c["uref"][None],
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(c["x0"])
    bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
    res = bt.ilqr_solve(iterations_max=40, tol_stationarity=1e-3)
    assert (res["status"] == 0).sum() >= batch * 0.98
    x, u = bt.get_nominal()
    sample = [0, 17, 333, 1023]
    refs = {}
    for b in sample:
        s = make_oracle(c, b)
        s.L.oracle_ilqr_set_options(s.h, 40, 1e-3, 1e-4, 1e-8, 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        np.testing.assert_allclose(x[b], s.get("x"), rtol=2e-7, atol=2e-7)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=2e-6, atol=2e-6)
        refs[b] = s
    # receding horizon: apply u_0, shift, re-solve warm-started
    for step in range(3):
        x1, _ = bt.get_knot(1)
        bt.set_initial_state(x1)
        bt.shift_trajectory()
        res = bt.ilqr_solve(iterations_max=40, tol_stationarity=1e-3)
        assert (res["status"] == 0).sum() >= batch * 0.98 and res["iterations"].mean() < 6
        xs, _ = bt.get_nominal()
        assert np.isfinite(xs).all()
        assert np.abs(np.linalg.norm(xs[:, :, 3:7], axis=2) - 1.0).max() < 0.05     # the attitude stays a (nearly) unit quaternion
    assert np.abs(xs[:, -1, :3]).mean() < np.abs(c["x0"][:, :3]).mean()             # the vehicles close in on the origin


def test_models_on_these_plans_say_what_they_take():
    bt = altro_amd.Batch(N, n, m, 4, plan=altro_amd.PLAN_GENERIC)
    with pytest.raises(altro_amd.AltroHipError, match="QUADROTOR13"):
        bt.set_model(altro_amd.MODEL_BICYCLE, H)
    bt = altro_amd.Batch(N, 14, 4, 4)
    with pytest.raises(altro_amd.AltroHipError, match="QUADROTOR13"):
        bt.set_model(altro_amd.MODEL_QUADROTOR13, H)
    bt = altro_amd.Batch(N, n, m, 4, dtype=altro_amd.F32)
    with pytest.raises(altro_amd.AltroHipError, match="fp64"):
        bt.set_model(altro_amd.MODEL_QUADROTOR13, H)


QUADROTOR13_SRC = r"""
// the caller's own 13-state quaternion quadrotor (the equations of csrc/models.h's MODEL_QUADROTOR13, written as a user would hand them over)
template <typename T>
__device__ void altro_user_dynamics(const T* x, const T* u, T* xd) {
  const T mass = T(0.5), g = T(9.81), Ix = T(0.0023), Iy = T(0.0023), Iz = T(0.004);
  const T qw = x[3], qx = x[4], qy = x[5], qz = x[6], wx = x[10], wy = x[11], wz = x[12];
  xd[0] = x[7]; xd[1] = x[8]; xd[2] = x[9];
  xd[3] = T(0.5) * (-qx * wx - qy * wy - qz * wz);
  xd[4] = T(0.5) * (qw * wx + qy * wz - qz * wy);
  xd[5] = T(0.5) * (qw * wy - qx * wz + qz * wx);
  xd[6] = T(0.5) * (qw * wz + qx * wy - qy * wx);
  const T a = u[0] * (T(1) / mass);
  xd[7] = a * (T(2) * (qx * qz + qw * qy));
  xd[8] = a * (T(2) * (qy * qz - qw * qx));
  xd[9] = a * (T(1) - T(2) * (qx * qx + qy * qy)) - g;
  xd[10] = (u[1] - (Iz - Iy) * wy * wz) * (T(1) / Ix);
  xd[11] = (u[2] - (Ix - Iz) * wz * wx) * (T(1) / Iy);
  xd[12] = (u[3] - (Iy - Ix) * wx * wy) * (T(1) / Iz);
}
template <typename T>
__device__ void altro_user_jacobian(const T* x, const T* u, T* J) {
  const int n = 13;
  const T mass = T(0.5), Ix = T(0.0023), Iy = T(0.0023), Iz = T(0.004);
  for (int e = 0; e < 13 * 17; ++e) J[e] = T(0);
  const T qw = x[3], qx = x[4], qy = x[5], qz = x[6], wx = x[10], wy = x[11], wz = x[12];
  J[0 + 7 * n] = T(1); J[1 + 8 * n] = T(1); J[2 + 9 * n] = T(1);
  J[3 + 4 * n] = T(-0.5) * wx; J[3 + 5 * n] = T(-0.5) * wy; J[3 + 6 * n] = T(-0.5) * wz; J[3 + 10 * n] = T(-0.5) * qx; J[3 + 11 * n] = T(-0.5) * qy; J[3 + 12 * n] = T(-0.5) * qz;
  J[4 + 3 * n] = T(0.5) * wx; J[4 + 5 * n] = T(0.5) * wz; J[4 + 6 * n] = T(-0.5) * wy; J[4 + 10 * n] = T(0.5) * qw; J[4 + 11 * n] = T(-0.5) * qz; J[4 + 12 * n] = T(0.5) * qy;
  J[5 + 3 * n] = T(0.5) * wy; J[5 + 4 * n] = T(-0.5) * wz; J[5 + 6 * n] = T(0.5) * wx; J[5 + 10 * n] = T(0.5) * qz; J[5 + 11 * n] = T(0.5) * qw; J[5 + 12 * n] = T(-0.5) * qx;
  J[6 + 3 * n] = T(0.5) * wz; J[6 + 4 * n] = T(0.5) * wy; J[6 + 5 * n] = T(-0.5) * wx; J[6 + 10 * n] = T(-0.5) * qy; J[6 + 11 * n] = T(0.5) * qx; J[6 + 12 * n] = T(0.5) * qw;
  const T rm = T(1) / mass, a = u[0] * rm;
  J[7 + 3 * n] = T(2) * a * qy; J[7 + 4 * n] = T(2) * a * qz; J[7 + 5 * n] = T(2) * a * qw; J[7 + 6 * n] = T(2) * a * qx;
  J[7 + 13 * n] = T(2) * (qx * qz + qw * qy) * rm;
  J[8 + 3 * n] = T(-2) * a * qx; J[8 + 4 * n] = T(-2) * a * qw; J[8 + 5 * n] = T(2) * a * qz; J[8 + 6 * n] = T(2) * a * qy;
  J[8 + 13 * n] = T(2) * (qy * qz - qw * qx) * rm;
  J[9 + 4 * n] = T(-4) * a * qx; J[9 + 5 * n] = T(-4) * a * qy;
  J[9 + 13 * n] = (T(1) - T(2) * (qx * qx + qy * qy)) * rm;
  J[10 + 11 * n] = -(Iz - Iy) * wz * (T(1) / Ix); J[10 + 12 * n] = -(Iz - Iy) * wy * (T(1) / Ix); J[10 + 14 * n] = T(1) / Ix;
  J[11 + 10 * n] = -(Ix - Iz) * wz * (T(1) / Iy); J[11 + 12 * n] = -(Ix - Iz) * wx * (T(1) / Iy); J[11 + 15 * n] = T(1) / Iy;
  J[12 + 10 * n] = -(Iy - Ix) * wy * (T(1) / Iz); J[12 + 11 * n] = -(Iy - Ix) * wx * (T(1) / Iz); J[12 + 16 * n] = T(1) / Iz;
}
"""


@pytest.mark.parametrize("plan", PLANS)
def test_a_callers_own_model_from_source_solves_like_the_compiled_in_one(plan):
    """altro_hip_set_model_source on plans GENERIC / MFMA32 (capi_rtc.hip compiles that plan's three model kernels around the caller's
    two function templates with hiprtc): the quaternion quadrotor handed over as source takes the compiled-in model's iterations and
    ends on its trajectory; with a thrust bound as an INEQUALITY block too."""
    batch = 12
    c = make_case(batch, seed=3)
    out = []
    for src in (False, True):
        bt = altro_amd.Batch(N, n, m, batch, plan=plan)
        if src:
            bt.set_model_source(QUADROTOR13_SRC, H)
        else:
            bt.set_model(altro_amd.MODEL_QUADROTOR13, H)
        # plan MFMA32: both models on the row-layout kernels (the source's Jacobian is written entry by entry: hiprtc keeps it in
        # registers, so its row-layout kernels are chosen -- altro_hip_model_row_layout); plan GENERIC by name: the same kernels
        assert bt.model_row_layout()
        bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]), c["Rd"][None], np.stack([c["xref"], c["xref"]]), c["uref"][None],
                             k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(c["x0"])
        bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
        Gb = np.zeros((2, n + m)); Gb[0, n] = 1.0; Gb[1, n] = -1.0
        bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.array([1.25 * HOVER[0], -0.6 * HOVER[0]]))
        res = bt.ilqr_solve(iterations_max=50, tol_stationarity=1e-3)
        x, u = bt.get_nominal()
        out.append((res, x, u))
    (ra, xa, ua), (rb, xb, ub) = out
    assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["iterations"], rb["iterations"])
    assert (ra["status"] == 0).sum() >= batch - 2
    np.testing.assert_allclose(xa, xb, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ua, ub, rtol=1e-8, atol=1e-8)


CHAIN_SRC = r"""
// sixteen states, five inputs: eight unit masses on a ring, coupled by hardening springs (force k d + c d^3), five of them actuated
template <typename T>
__device__ void altro_user_dynamics(const T* x, const T* u, T* xd) {
  for (int i = 0; i < 8; ++i) {
    const int l = (i + 7) % 8, r = (i + 1) % 8;
    const T dl = x[l] - x[i], dr = x[r] - x[i];
    xd[i] = x[8 + i];
    xd[8 + i] = T(2) * (dl + dr) + T(0.5) * (dl * dl * dl + dr * dr * dr) - T(0.1) * x[8 + i] + (i < 5 ? u[i] : T(0));
  }
}
template <typename T>
__device__ void altro_user_jacobian(const T* x, const T* u, T* J) {
  const int n = 16;
  for (int e = 0; e < 16 * 21; ++e) J[e] = T(0);
  for (int i = 0; i < 8; ++i) {
    const int l = (i + 7) % 8, r = (i + 1) % 8;
    const T dl = x[l] - x[i], dr = x[r] - x[i];
    const T kl = T(2) + T(1.5) * dl * dl, kr = T(2) + T(1.5) * dr * dr;
    J[i + (8 + i) * n] = T(1);
    J[(8 + i) + l * n] += kl; J[(8 + i) + r * n] += kr; J[(8 + i) + i * n] += -(kl + kr);
    J[(8 + i) + (8 + i) * n] = T(-0.1);
    if (i < 5) J[(8 + i) + (16 + i) * n] = T(1);
  }
}
"""


def test_a_sixteen_state_model_from_source():
    """A shape no compiled-in model has, (16, 5): rollout against numpy's explicit midpoint rule, A_k, B_k against central differences of
    that, and whole solves that converge (the reference takes any (n, m) through SetExplicitDynamics: altro_solver.cpp:68-81)."""
    nn, mm, NN, batch = 16, 5, 25, 40
    h = np.float32(0.05)

    def f(x, u):
        xd = np.zeros_like(x)
        for i in range(8):
            l, r = (i + 7) % 8, (i + 1) % 8
            dl, dr = x[..., l] - x[..., i], x[..., r] - x[..., i]
            xd[..., i] = x[..., 8 + i]
            xd[..., 8 + i] = 2 * (dl + dr) + 0.5 * (dl ** 3 + dr ** 3) - 0.1 * x[..., 8 + i] + (u[..., i] if i < 5 else 0.0)
        return xd

    def step(x, u):
        return x + float(h) * f(x + float(np.float32(h / 2)) * f(x, u), u)
    rng = np.random.default_rng(5)
    x0 = 0.5 * rng.standard_normal((batch, nn))
    u0 = 0.3 * rng.standard_normal((batch, NN, mm))
    bt = altro_amd.Batch(NN, nn, mm, batch)
    assert bt.plan == altro_amd.PLAN_MFMA32
    bt.set_model_source(CHAIN_SRC, h)
    bt.set_tracking_cost(np.stack([np.ones(nn), 10.0 * np.ones(nn)]), np.full((1, mm), 0.1), np.zeros((2, nn)), np.zeros((1, mm)),
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0); bt.set_input_guess(u0)
    bt.open_loop_rollout(); bt.accept(); bt.expand()
    xr = bt.get("x").reshape(batch, NN + 1, nn)
    x = x0.copy()
    for k in range(NN):
        np.testing.assert_allclose(xr[:, k], x, rtol=1e-11, atol=1e-11)
        x = step(x, u0[:, k])
    A, B, _, _ = bt.get_expansion()
    for b in (0, 17, 39):
        for k in (0, 11, NN - 1):
            Ak, Bk = A[b, k].reshape(nn, nn).T, B[b, k].reshape(mm, nn).T
            for c in range(nn + mm):
                e = np.zeros(nn + mm); e[c] = 1e-6
                fd = (step(xr[b, k] + e[:nn], u0[b, k] + e[nn:]) - step(xr[b, k] - e[:nn], u0[b, k] - e[nn:])) / 2e-6
                col = Ak[:, c] if c < nn else Bk[:, c - nn]
                assert np.abs(fd - col).max() < 1e-6, (b, k, c)
    res = bt.ilqr_solve(iterations_max=60, tol_stationarity=1e-3)
    assert (res["status"] == 0).sum() >= batch - 2 and res["iterations"].max() < 60
    # whichever model kernels the source got (the row layout when hiprtc kept its Jacobian in registers: altro_hip_model_row_layout),
    # the wave-per-problem form takes the same iterations to the same trajectory
    print("the (16, 5) chain from source runs the %s model kernels" % ("row-layout" if bt.model_row_layout() else "wave-per-problem"))
    b2 = altro_amd.Batch(NN, nn, mm, batch)
    b2.set_forms(altro_amd.FORM_GENERIC_MERIT_LDS)
    b2.set_model_source(CHAIN_SRC, h)
    assert not b2.model_row_layout()
    b2.set_tracking_cost(np.stack([np.ones(nn), 10.0 * np.ones(nn)]), np.full((1, mm), 0.1), np.zeros((2, nn)), np.zeros((1, mm)),
                         k_stride_zero=True, batch_stride_zero=True)
    b2.set_initial_state(x0); b2.set_input_guess(u0)
    r2 = b2.ilqr_solve(iterations_max=60, tol_stationarity=1e-3)
    assert np.array_equal(res["status"], r2["status"]) and np.array_equal(res["iterations"], r2["iterations"])
    np.testing.assert_allclose(bt.get_nominal()[0], b2.get_nominal()[0], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("constrained", [False, True])
def test_row_layout_model_merit_equals_the_lds_form(constrained):
    """MeritFunction with the compiled-in model in the row layout (kernels/ilqr_row32.hip: r32_model_step -- every lane evaluates the
    model, two problems per wave) against generic_merit_kernel<.., MK> (lane 0 evaluates it for a wave of one problem:
    ALTRO_HIP_FORM_GENERIC_MERIT_LDS): the same sums in the same order -- the first expansion's A_k, B_k, phi, phi', the candidate, the refreshed A_k, B_k, the
    stored gradient at 1e-13 relative (the printout says whether they are bit-identical), with and without the derivative, odd batch;
    whole solves with the same status and iteration count."""
    batch = 7
    c = make_case(batch, seed=3)
    Gb = np.zeros((2, n + m)); Gb[0, n] = 1.0; Gb[1, n] = -1.0
    blocks = [(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.array([1.1 * HOVER[0], -0.8 * HOVER[0]]))] if constrained else []
    out = {}
    for name, forms in (("row", 0), ("lds", altro_amd.FORM_GENERIC_MERIT_LDS)):
        bt = make_hip(c, altro_amd.PLAN_MFMA32)
        bt.set_forms(forms)
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
        bt.open_loop_rollout(); bt.accept(); bt.expand()
        r = {}
        r["A_expand"], r["B_expand"] = (v.copy() for v in bt.get_expansion()[:2])   # (row32_expand_dyn_kernel / generic_model_expand_dyn_kernel)
        bt.backward()
        assert (bt.get("status") == -1).all()
        for tag, alphas, deriv in (("a", np.linspace(0.05, 1.0, batch), True), ("zero", np.zeros(batch), True), ("noderiv", np.full(batch, 0.5), False)):
            phi, dphi = bt.merit(alphas, derivative=deriv)
            r["phi_" + tag] = phi.copy()
            r["x_" + tag] = bt.get("x").copy(); r["u_" + tag] = bt.get("u").copy(); r["y_" + tag] = bt.get("y").copy()
            if deriv:
                r["dphi_" + tag] = dphi.copy()
                A, B, lx, lu = bt.get_expansion()
                r["A_" + tag] = A.copy(); r["B_" + tag] = B.copy(); r["lx_" + tag] = lx.copy(); r["lu_" + tag] = lu.copy()
                r["stat_" + tag] = np.asarray(bt.stationarity(), dtype=np.float64).copy()
        res = bt.ilqr_solve(iterations_max=50, tol_stationarity=1e-3, penalty_initial=1.0, penalty_scaling=10.0)
        r["solve_status"] = res["status"].copy(); r["solve_iterations"] = res["iterations"].copy()
        r["solve_x"] = bt.get_nominal()[0].copy()
        out[name] = r
        bt.close()
    exact = 0
    for key in out["row"]:
        a, b = out["row"][key], out["lds"][key]
        if key in ("solve_status", "solve_iterations"):
            assert np.array_equal(a, b), (key, a, b)
        elif key == "solve_x":
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(b).max())), err_msg=key)
        exact += int(np.array_equal(a, b))
    assert (out["row"]["solve_status"] == 0).sum() >= batch - 2
    print("row-layout model merit == LDS form bit for bit in %d of %d quantities%s" % (exact, len(out["row"]), " (thrust bound)" if constrained else ""))


DENSE_SRC = r"""
// fourteen states, five inputs, every state coupled to every other: xdot_i = -x_i + 0.1 sum_j sin(x_j - x_i) + (i < 5 ? u_i : 0)
template <typename T>
__device__ void altro_user_dynamics(const T* x, const T* u, T* xd) {
  for (int i = 0; i < 14; ++i) {
    T s = T(0);
    for (int j = 0; j < 14; ++j) s += sin(x[j] - x[i]);
    xd[i] = -x[i] + T(0.1) * s + (i < 5 ? u[i] : T(0));
  }
}
template <typename T>
__device__ void altro_user_jacobian(const T* x, const T* u, T* J) {
  const int n = 14;
  for (int e = 0; e < 14 * 19; ++e) J[e] = T(0);
  for (int i = 0; i < 14; ++i) {
    T d = T(0);
    for (int j = 0; j < 14; ++j) {
      const T c = T(0.1) * cos(x[j] - x[i]);
      if (j != i) { J[i + j * n] = c; d += c; }
    }
    J[i + i * n] = T(-1) - d;
    if (i < 5) J[i + (14 + i) * n] = T(1);
  }
}
"""


def test_a_dense_model_from_source_keeps_the_wave_per_problem_kernels():
    """The row-layout model kernels keep the whole first Jacobian in every lane's registers; a DENSE one of fourteen states does not fit
    (hiprtc's kernels for it use scratch memory), so altro_hip_set_model_source leaves such a source on the wave-per-problem kernels --
    altro_hip_model_row_layout says so -- and the solve is what it was: A_k, B_k against central differences of numpy's midpoint rule,
    whole solves converge."""
    nn, mm, NN, batch = 14, 5, 20, 12
    h = np.float32(0.05)

    def f(x, u):
        d = x[..., None, :] - x[..., :, None]                       # d[i, j] = x_j - x_i
        xd = -x + 0.1 * np.sin(d).sum(axis=-1)
        xd[..., :5] += u
        return xd

    def step(x, u):
        return x + float(h) * f(x + float(np.float32(h / 2)) * f(x, u), u)
    rng = np.random.default_rng(11)
    x0 = 0.6 * rng.standard_normal((batch, nn)); u0 = 0.2 * rng.standard_normal((batch, NN, mm))
    bt = altro_amd.Batch(NN, nn, mm, batch)
    assert bt.plan == altro_amd.PLAN_MFMA32
    bt.set_model_source(DENSE_SRC, h)
    assert not bt.model_row_layout()
    bt.set_tracking_cost(np.stack([np.ones(nn), 10.0 * np.ones(nn)]), np.full((1, mm), 0.1), np.zeros((2, nn)), np.zeros((1, mm)),
                         k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0); bt.set_input_guess(u0)
    bt.open_loop_rollout(); bt.accept(); bt.expand()
    xr = bt.get("x").reshape(batch, NN + 1, nn)
    A, B, _, _ = bt.get_expansion()
    for b in (0, 7):
        for k in (0, NN - 1):
            Ak, Bk = A[b, k].reshape(nn, nn).T, B[b, k].reshape(mm, nn).T
            for c in range(nn + mm):
                e = np.zeros(nn + mm); e[c] = 1e-6
                fd = (step(xr[b, k] + e[:nn], u0[b, k] + e[nn:]) - step(xr[b, k] - e[:nn], u0[b, k] - e[nn:])) / 2e-6
                col = Ak[:, c] if c < nn else Bk[:, c - nn]
                assert np.abs(fd - col).max() < 1e-6, (b, k, c)
    res = bt.ilqr_solve(iterations_max=60, tol_stationarity=1e-3)
    assert (res["status"] == 0).sum() >= batch - 1
    bt.close()
