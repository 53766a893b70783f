"""NONLINEAR dynamics on the (12, 4) tile plan (missing #2): altro_hip_set_model on plan MFMA16 -- the rollout and the
merit evaluation step a device model (explicit midpoint rule, test/test_utils.cpp:84-132), the expansion leaves A, B in the
records the backward sweep reads (KnotPointData::CalcDynamicsExpansion, knotpoint_data.cpp:406-419) -- against the oracle running
the same model through its restatement of SolverImpl.  The model is the 12-state quadrotor of csrc/models.h / oracle/models_oracle.c
(this repo's own test model: the reference ships none that large; its Jacobian is pinned by central differences in
tests/test_oracle_kat.py).

Tolerances: one merit evaluation phi 1e-10, phi' 1e-8 relative, candidates 2e-9 / 2e-8 (device vs glibc sin / cos differ in the
last ulp, the tile's sums are reassociated, and 40 closed-loop steps of a quadrotor with gains of order 1e2 amplify both: the first
run measured 1.7e-10 on one entry of 492); whole solves: same status and iterations per problem, trajectories 1e-6."""
import numpy as np
import pytest

import altro_amd
from oracle import oracle
from tests import problems

pytestmark = pytest.mark.gpu

N, n, m = 40, 12, 4
H = np.float32(0.02)
HOVER = np.array([0.5 * 9.81, 0.0, 0.0, 0.0])


def make_case(batch, dense):
    """Fly from a perturbed state to hover at the origin: tracking cost (or the same as dense blocks + a cross term)."""
    x0 = np.zeros((batch, n))
    x0[:, :3] = 0.8 * problems.normal((batch, 3), 91)
    x0[:, 3:6] = 0.15 * problems.normal((batch, 3), 92)
    x0[:, 6:9] = 0.3 * problems.normal((batch, 3), 93)
    x0[:, 9:] = 0.2 * problems.normal((batch, 3), 94)
    Qd = np.concatenate([np.full(3, 2.0), np.full(3, 1.0), np.full(3, 0.5), np.full(3, 0.1)])
    Rd = np.array([0.05, 20.0, 20.0, 20.0])
    c = dict(x0=x0, Qd=Qd, Qfd=20.0 * Qd, Rd=Rd, xref=np.zeros(n), uref=HOVER, u0=HOVER.copy())
    if dense:
        c.update(problems.quadratic_cost(batch, N, n, m, stream=121))
        c["R"] = c["R"] + np.diag([0.0, 20.0, 20.0, 20.0]).T.reshape(-1)     # torques are tiny numbers: weigh them like the tracking cost
        c["r"] = c["r"] * 0.0 - (c["R"].reshape(batch, N, m, m) @ HOVER)      # minimum at the hover input
        c["q"] = c["q"] * 0.1
    return c


def make_hip(c, dense):
    bt = altro_amd.Batch(N, n, m, c["x0"].shape[0])
    assert bt.plan == altro_amd.PLAN_MFMA16
    bt.set_model(altro_amd.MODEL_QUADROTOR, H)
    if dense:
        bt.set_quadratic_cost(c["Q"], c["R"], c["H"], c["q"], c["r"], c["c"])
    else:
        bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]), c["Rd"][None], np.stack([c["xref"], c["xref"]]), c["uref"][None],
                             k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(c["x0"])
    bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
    return bt


def make_oracle(c, b, dense, blocks=()):
    s = oracle.ILQR(N, n, m, H, oracle.DYN_MODEL, oracle.MODEL_QUADROTOR, cost_kind=oracle.COST_QUADRATIC if dense else oracle.COST_DIAGONAL)
    for k in range(N + 1):
        kk = min(k, N - 1)
        if dense:
            s.L.oracle_ilqr_set_quadratic_cost(s.h, k, np.ascontiguousarray(c["Q"][b, k]), np.ascontiguousarray(c["R"][b, kk]).ctypes.data,
                                               np.ascontiguousarray(c["H"][b, kk]).ctypes.data, np.ascontiguousarray(c["q"][b, k]),
                                               np.ascontiguousarray(c["r"][b, kk]).ctypes.data, float(c["c"][b, k]))
        else:
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


@pytest.mark.parametrize("dense", [False, True])
def test_merit_expansion_stationarity(dense):
    """Rollout, expansion (A, B through the gains of the backward sweep), one merit evaluation per problem at its own alpha:
    phi, phi', candidate x_ / u_ / y_, and the candidate's stationarity (which reads the Z rows the pass left in the records)."""
    batch = 9
    c = make_case(batch, dense)
    bt = make_hip(c, dense)
    bt.open_loop_rollout(); bt.accept(); bt.expand()
    A0, B0, lx0, lu0 = bt.get_expansion()             # the expansion at the rolled-out trajectory (before any merit pass)
    bt.backward()
    assert (bt.get("status") == -1).all()
    alphas = np.linspace(0.0, 1.0, batch)
    phi, dphi = bt.merit(alphas)
    xc, uc, yc = bt.get("x"), bt.get("u"), bt.get("y")
    st = bt.stationarity()
    K = bt.get("K")
    for b in [0, 3, 8]:
        s = make_oracle(c, b, dense)
        s.L.oracle_ilqr_open_loop_rollout(s.h); s.L.oracle_ilqr_copy_trajectory(s.h)
        s.L.oracle_ilqr_calc_dynamics_expansions(s.h); s.L.oracle_ilqr_calc_cost_gradient(s.h)
        s.L.oracle_ilqr_calc_expansions(s.h)
        np.testing.assert_allclose(A0[b], s.get("A"), rtol=1e-12, atol=1e-12)    # CalcDynamicsExpansion: A, B of the midpoint rule
        np.testing.assert_allclose(B0[b], s.get("B"), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(lx0[b], s.get("lx"), rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(lu0[b], s.get("lu"), rtol=1e-11, atol=1e-11)
        assert s.L.oracle_ilqr_backward_pass(s.h) == -1
        Kr = s.get("K")
        assert np.abs(K[b] - Kr).max() <= 1e-8 * max(1.0, np.abs(Kr).max())          # gains: A, B, lxx, luu, lux all entered
        p_ref, dp_ref = s.merit(alphas[b])
        assert abs(phi[b] - p_ref) <= 1e-10 * max(1.0, abs(p_ref)), (b, phi[b], p_ref)
        assert abs(dphi[b] - dp_ref) <= 1e-8 * max(1.0, abs(dp_ref)), (b, dphi[b], dp_ref)
        np.testing.assert_allclose(xc[b], s.get("x_cand"), rtol=2e-9, atol=2e-9)
        np.testing.assert_allclose(uc[b], s.get("u_cand"), rtol=2e-8, atol=2e-8)
        np.testing.assert_allclose(yc[b], s.get("y_cand"), rtol=1e-7, atol=1e-7)
        ref_st = s.L.oracle_ilqr_stationarity(s.h)
        assert abs(st[b] - ref_st) <= 1e-7 * max(1.0, ref_st), (b, st[b], ref_st)


@pytest.mark.parametrize("dense,constrained,backtracking", [(False, False, False), (False, True, False), (True, False, False),
                                                             (False, False, True), (True, True, False)])
def test_whole_solves(dense, constrained, backtracking):
    """Whole (AL-)iLQR solves of the nonlinear 12-state problem on the tile plan: status, iterations, trajectories per problem;
    with thrust / torque bounds as an INEQUALITY block when `constrained`."""
    batch = 21
    c = make_case(batch, dense)
    bt = make_hip(c, dense)
    blocks = []
    if constrained:
        Gb = np.zeros((2, n + m)); Gb[0, 12] = 1.0; Gb[1, 12] = -1.0
        blocks = [(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.array([1.25 * HOVER[0], -0.6 * HOVER[0]]))]
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
    # (tol_stationarity 1e-3: at the default 1e-4 most of these stall just above it -- phi' below tol_meritfun_gradient, step 0,
    #  the reference's own behaviour -- and run into the iteration limit, device and oracle alike)
    res = bt.ilqr_solve(iterations_max=50, use_backtracking=backtracking, tol_stationarity=1e-3)
    x, u = bt.get_nominal()
    nconv = 0
    for b in [0, 10, 20]:
        s = make_oracle(c, b, dense, blocks)
        if blocks:
            s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 50, 1e-3, 1e-4, 1e-8, 1 if backtracking else 0)
        status, iters, log = s.solve()
        assert res["status"][b] == status and res["iterations"][b] == iters, (b, res["status"][b], status, res["iterations"][b], iters)
        if status != 0:
            continue
        nconv += 1
        np.testing.assert_allclose(x[b], s.get("x"), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(u[b], s.get("u"), rtol=1e-5, atol=1e-5)
    assert nconv >= 2


def test_model_needs_the_tile_shape_and_fp64():
    bt = altro_amd.Batch(N, 9, 3, 4, plan=altro_amd.PLAN_MFMA16)
    with pytest.raises(altro_amd.AltroHipError):
        bt.set_model(altro_amd.MODEL_QUADROTOR, H)
    bt = altro_amd.Batch(N, 12, 4, 4, dtype=altro_amd.F32)
    with pytest.raises(altro_amd.AltroHipError):
        bt.set_model(altro_amd.MODEL_QUADROTOR, H)


QUADROTOR_SRC = r"""
// the caller's own 12-state quadrotor (the equations of csrc/models.h's MODEL_QUADROTOR, written as a user would hand them over)
template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xd) {
  const T mass = T(0.5), g = T(9.81), Ix = T(0.0023), Iy = T(0.0023), Iz = T(0.004);
  T sp, cp, st, ct, ss, cs;
  sincos(x[3], &sp, &cp); sincos(x[4], &st, &ct); sincos(x[5], &ss, &cs);
  const T tt = st / ct, wx = x[9], wy = x[10], wz = x[11];
  xd[0] = x[6]; xd[1] = x[7]; xd[2] = x[8];
  xd[3] = wx + sp * tt * wy + cp * tt * wz;
  xd[4] = cp * wy - sp * wz;
  xd[5] = (sp * wy + cp * wz) / ct;
  const T a = u[0] / mass;
  xd[6] = a * (cp * st * cs + sp * ss);
  xd[7] = a * (cp * st * ss - sp * cs);
  xd[8] = a * (cp * ct) - g;
  xd[9] = (u[1] - (Iz - Iy) * wy * wz) / Ix;
  xd[10] = (u[2] - (Ix - Iz) * wz * wx) / Iy;
  xd[11] = (u[3] - (Iy - Ix) * wx * wy) / Iz;
}
template <typename T> __device__ void altro_user_jacobian(const T* x, const T* u, T* J) {
  const T mass = T(0.5), Ix = T(0.0023), Iy = T(0.0023), Iz = T(0.004);
  T sp, cp, st, ct, ss, cs;
  sincos(x[3], &sp, &cp); sincos(x[4], &st, &ct); sincos(x[5], &ss, &cs);
  const T tt = st / ct, sec2 = T(1) / (ct * ct), wx = x[9], wy = x[10], wz = x[11];
  for (int e = 0; e < 192; ++e) J[e] = T(0);
  J[0 + 6 * 12] = T(1); J[1 + 7 * 12] = T(1); J[2 + 8 * 12] = T(1);
  J[3 + 3 * 12] = cp * tt * wy - sp * tt * wz; J[3 + 4 * 12] = (sp * wy + cp * wz) * sec2;
  J[3 + 9 * 12] = T(1); J[3 + 10 * 12] = sp * tt; J[3 + 11 * 12] = cp * tt;
  J[4 + 3 * 12] = -sp * wy - cp * wz; J[4 + 10 * 12] = cp; J[4 + 11 * 12] = -sp;
  J[5 + 3 * 12] = (cp * wy - sp * wz) / ct; J[5 + 4 * 12] = (sp * wy + cp * wz) * st * sec2;
  J[5 + 10 * 12] = sp / ct; J[5 + 11 * 12] = cp / ct;
  const T a = u[0] / mass;
  J[6 + 3 * 12] = a * (-sp * st * cs + cp * ss); J[6 + 4 * 12] = a * (cp * ct * cs); J[6 + 5 * 12] = a * (-cp * st * ss + sp * cs);
  J[6 + 12 * 12] = (cp * st * cs + sp * ss) / mass;
  J[7 + 3 * 12] = a * (-sp * st * ss - cp * cs); J[7 + 4 * 12] = a * (cp * ct * ss); J[7 + 5 * 12] = a * (cp * st * cs + sp * ss);
  J[7 + 12 * 12] = (cp * st * ss - sp * cs) / mass;
  J[8 + 3 * 12] = a * (-sp * ct); J[8 + 4 * 12] = a * (-cp * st); J[8 + 12 * 12] = (cp * ct) / mass;
  J[9 + 10 * 12] = -(Iz - Iy) * wz / Ix; J[9 + 11 * 12] = -(Iz - Iy) * wy / Ix; J[9 + 13 * 12] = T(1) / Ix;
  J[10 + 9 * 12] = -(Ix - Iz) * wz / Iy; J[10 + 11 * 12] = -(Ix - Iz) * wx / Iy; J[10 + 14 * 12] = T(1) / Iy;
  J[11 + 9 * 12] = -(Iy - Ix) * wy / Iz; J[11 + 10 * 12] = -(Iy - Ix) * wx / Iz; J[11 + 15 * 12] = T(1) / Iz;
}
"""


@pytest.mark.parametrize("constrained", [False, True])
def test_user_source_on_the_tile_plan(constrained):
    """altro_hip_set_model_source on plan MFMA16: the caller's own 12-state dynamics + Jacobian as HIP source, compiled by hiprtc into
    the tile plan's row-layout kernels.  The quadrotor handed over as source solves like the compiled-in MODEL_QUADROTOR (same
    equations, same kernels around them: status, iterations, trajectories 1e-10) and like the oracle."""
    batch = 13
    c = make_case(batch, False)
    blocks = []
    if constrained:
        Gb = np.zeros((2, n + m)); Gb[0, 12] = 1.0; Gb[1, 12] = -1.0
        blocks = [(0, N - 1, altro_amd.CONE_INEQUALITY, Gb, np.array([1.25 * HOVER[0], -0.6 * HOVER[0]]))]

    def run(source):
        bt = altro_amd.Batch(N, n, m, batch)
        for (k0, k1, cone, G, g) in blocks:
            bt.add_linear_constraint(k0, k1, cone, G, g)
        if source:
            bt.set_model_source(QUADROTOR_SRC, H)
        else:
            bt.set_model(altro_amd.MODEL_QUADROTOR, H)
        bt.set_tracking_cost(np.stack([c["Qd"], c["Qfd"]]), c["Rd"][None], np.stack([c["xref"], c["xref"]]), c["uref"][None],
                             k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(c["x0"])
        bt.set_input_guess(c["u0"][None, None], k_stride_zero=True, batch_stride_zero=True)
        res = bt.ilqr_solve(iterations_max=50, tol_stationarity=1e-3)
        return res, bt.get_nominal()
    r_src, (x_src, u_src) = run(True)
    r_mod, (x_mod, u_mod) = run(False)
    assert np.array_equal(r_src["status"], r_mod["status"]) and np.array_equal(r_src["iterations"], r_mod["iterations"])
    np.testing.assert_allclose(x_src, x_mod, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(u_src, u_mod, rtol=1e-9, atol=1e-9)
    s = make_oracle(c, 0, False, blocks)      # (problem 0: one of those test_whole_solves holds against the oracle)
    if blocks:
        s.set_penalty(1.0, 10.0)
    s.L.oracle_ilqr_set_options(s.h, 50, 1e-3, 1e-4, 1e-8, 0)
    status, iters, log = s.solve()
    assert r_src["status"][0] == status and r_src["iterations"][0] == iters
    np.testing.assert_allclose(x_src[0], s.get("x"), rtol=1e-6, atol=1e-6)


PLANAR_SRC = r"""
// planar quadrotor: x = [px, pz, theta, vx, vz, omega], u = [left thrust, right thrust]
template <typename T> __device__ void altro_user_dynamics(const T* x, const T* u, T* xd) {
  const T mass = T(1.0), g = T(9.81), l = T(0.25), I = T(0.05);
  const T s = sin(x[2]), c = cos(x[2]), F = u[0] + u[1];
  xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
  xd[3] = -F * s / mass; xd[4] = F * c / mass - g; xd[5] = l * (u[1] - u[0]) / I;
}
template <typename T> __device__ void altro_user_jacobian(const T* x, const T* u, T* J) {
  const T mass = T(1.0), l = T(0.25), I = T(0.05);
  const T s = sin(x[2]), c = cos(x[2]), F = u[0] + u[1];
  for (int e = 0; e < 48; ++e) J[e] = T(0);
  J[0 + 3 * 6] = T(1); J[1 + 4 * 6] = T(1); J[2 + 5 * 6] = T(1);
  J[3 + 2 * 6] = -F * c / mass; J[4 + 2 * 6] = -F * s / mass;
  J[3 + 6 * 6] = -s / mass; J[3 + 7 * 6] = -s / mass;
  J[4 + 6 * 6] = c / mass; J[4 + 7 * 6] = c / mass;
  J[5 + 6 * 6] = -l / I; J[5 + 7 * 6] = l / I;
}
"""


def test_padded_user_model_tile_plan_equals_plan_lane():
    """A caller's (6, 2) model rides the (12, 4) tile zero-padded (the shim of models.h re-lays its Jacobian into the tile's
    columns): the same source solved on plan MFMA16 and on plan LANE (lane-per-problem kernels, compared with the oracle in
    tests/test_gpu_user_model.py) gives the same iterations and trajectories."""
    batch, Np, nn, mm = 17, 30, 6, 2
    hp = np.float32(0.05)
    x0 = np.zeros((batch, nn)); x0[:, :2] = 0.6 * problems.normal((batch, 2), 141); x0[:, 2] = 0.2 * problems.normal((batch,), 142)
    Qd = np.array([2.0, 2.0, 1.0, 0.3, 0.3, 0.1]); Rd = np.array([0.1, 0.1]); uh = np.full(2, 0.5 * 9.81)
    out = []
    for plan in (altro_amd.PLAN_MFMA16, altro_amd.PLAN_LANE):
        bt = altro_amd.Batch(Np, nn, mm, batch, plan=plan)
        assert bt.plan == plan
        bt.set_model_source(PLANAR_SRC, hp)
        bt.set_tracking_cost(np.stack([Qd, 30.0 * Qd]), Rd[None], np.zeros((2, nn)), uh[None], k_stride_zero=True, batch_stride_zero=True)
        bt.set_initial_state(x0)
        bt.set_input_guess(uh[None, None], k_stride_zero=True, batch_stride_zero=True)
        res = bt.ilqr_solve(iterations_max=60, tol_stationarity=1e-3)
        x, u = bt.get_nominal()
        assert x.shape == (batch, Np + 1, nn) and u.shape == (batch, Np, mm)
        out.append((res, x, u))
    (ra, xa, ua), (rb, xb, ub) = out
    assert (ra["status"] == 0).sum() >= batch - 2
    same = ra["iterations"] == rb["iterations"]
    assert same.sum() >= batch - 1, (ra["iterations"], rb["iterations"])     # (a convergence test at rounding level may move by one sweep)
    np.testing.assert_allclose(xa[same], xb[same], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(ua[same], ub[same], rtol=1e-6, atol=1e-6)


def test_full_size_quadrotor_rollout_and_expansion():
    """The tile plan's device model at BASELINE.json configs[1]'s size (4096 problems, N = 256): size-independent checks.  The open-loop
    rollout of every problem is finite and, for a seeded sample, equals the explicit-midpoint integration of the same equations in numpy
    (1e-11); the expansion's A_k, B_k equal central differences of that numpy step (1e-6) at sampled knot points."""
    batch, Nf = 4096, 256
    mass, grav, ix, iy, iz = 0.5, 9.81, 0.0023, 0.0023, 0.004

    def f_cont(x, u):
        sp, cp, st, ct, ss, cs = np.sin(x[3]), np.cos(x[3]), np.sin(x[4]), np.cos(x[4]), np.sin(x[5]), np.cos(x[5])
        tt = st / ct
        wx, wy, wz = x[9], x[10], x[11]
        a = u[0] / mass
        return np.array([x[6], x[7], x[8], wx + sp * tt * wy + cp * tt * wz, cp * wy - sp * wz, (sp * wy + cp * wz) / ct,
                         a * (cp * st * cs + sp * ss), a * (cp * st * ss - sp * cs), a * (cp * ct) - grav,
                         (u[1] - (iz - iy) * wy * wz) / ix, (u[2] - (ix - iz) * wz * wx) / iy, (u[3] - (iy - ix) * wx * wy) / iz])

    hd = float(np.float32(0.01)); h2 = float(np.float32(0.01) / np.float32(2))
    step = lambda x, u: x + hd * f_cont(x + h2 * f_cont(x, u), u)   # noqa: E731
    x0 = np.zeros((batch, n))
    x0[:, :3] = 0.5 * problems.normal((batch, 3), 191)
    x0[:, 3:6] = 0.1 * problems.normal((batch, 3), 192)
    x0[:, 6:9] = 0.2 * problems.normal((batch, 3), 193)
    x0[:, 9:] = 0.1 * problems.normal((batch, 3), 194)
    u = HOVER[None, None] + np.array([0.3, 0.002, 0.002, 0.002]) * problems.normal((batch, Nf, m), 195)
    bt = altro_amd.Batch(Nf, n, m, batch)
    bt.set_model(altro_amd.MODEL_QUADROTOR, np.float32(0.01))
    Qd = np.ones(n)
    bt.set_tracking_cost(np.stack([Qd, Qd]), np.ones((1, m)), np.zeros((2, n)), HOVER[None], k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0)
    bt.set_input_guess(u)
    bt.open_loop_rollout(); bt.accept(); bt.expand()
    x, _ = bt.get_nominal()
    assert np.isfinite(x).all()
    A, B, _, _ = bt.get_expansion()
    for b in (0, 1234, 4095):
        xr = np.zeros((Nf + 1, n)); xr[0] = x0[b]
        for k in range(Nf):
            xr[k + 1] = step(xr[k], u[b, k])
        np.testing.assert_allclose(x[b], xr, rtol=1e-11, atol=1e-11)
        for k in (0, 100, 255):
            Ak = A[b, k].reshape(n, n).T; Bk = B[b, k].reshape(m, n).T
            eps = 1e-6
            for j in range(n):
                e = np.zeros(n); e[j] = eps
                np.testing.assert_allclose(Ak[:, j], (step(xr[k] + e, u[b, k]) - step(xr[k] - e, u[b, k])) / (2 * eps), rtol=1e-6, atol=1e-6)
            for j in range(m):
                e = np.zeros(m); e[j] = eps
                np.testing.assert_allclose(Bk[:, j], (step(xr[k], u[b, k] + e) - step(xr[k], u[b, k] - e)) / (2 * eps), rtol=1e-6, atol=1e-6)
    bt.close()
