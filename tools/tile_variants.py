"""tools/tile_variants.py -- what the round-4 instantiations of the (12, 4) tile plan's kernels cost, on one box, one process:
4096 problems, N = 64; {dynamics as data, quadrotor model} x {diagonal, dense cost} x {free, |u - u_hover| <= bound}.
Per variant: the median wall time of one merit evaluation with derivative (altro_hip_merit: the pass that also stores the next
expansion), of one backward sweep, and of a whole altro_hip_ilqr_solve.  Host clock around synchronised calls (the calls block).
Usage: python tools/tile_variants.py [batch] [N]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n, m = 12, 4
H = np.float32(0.02)
HOVER = np.array([0.5 * 9.81, 0.0, 0.0, 0.0])


def med(f, reps=15):
    f(); f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


def build(model, dense, bounded):
    bt = altro_amd.Batch(N, n, m, batch)
    assert bt.plan == altro_amd.PLAN_MFMA16
    x0 = np.zeros((batch, n))
    x0[:, :3] = 0.8 * problems.normal((batch, 3), 91)
    x0[:, 3:6] = 0.15 * problems.normal((batch, 3), 92)
    x0[:, 6:9] = 0.3 * problems.normal((batch, 3), 93)
    x0[:, 9:] = 0.2 * problems.normal((batch, 3), 94)
    if model:
        bt.set_model(altro_amd.MODEL_QUADROTOR, H)
    else:   # the same vehicle linearised at hover, as data shared by every (problem, knot point) (expanded on the device)
        lt = problems.quadrotor_ltv(1, 1, h=0.02)
        bt.set_dynamics(lt["A"][0, :1], lt["B"][0, :1], None, k_stride_zero=True, batch_stride_zero=True)
    Qd = np.concatenate([np.full(3, 2.0), np.full(3, 1.0), np.full(3, 0.5), np.full(3, 0.1)])
    Rd = np.array([0.05, 20.0, 20.0, 20.0])
    uref = HOVER if model else np.zeros(m)
    if dense:
        c = problems.quadratic_cost(1, N, n, m, stream=121)
        R = c["R"] + np.diag([0.0, 20.0, 20.0, 20.0]).reshape(-1)
        r = -(R.reshape(1, N, m, m) @ uref)
        bt.set_quadratic_cost(c["Q"], R, c["H"], 0.1 * c["q"], r, c["c"], batch_stride_zero=True)
    else:
        bt.set_tracking_cost(np.stack([Qd, 20.0 * Qd]), Rd[None], np.zeros((2, n)), uref[None], k_stride_zero=True, batch_stride_zero=True)
    bt.set_initial_state(x0)
    bt.set_input_guess(uref[None, None], k_stride_zero=True, batch_stride_zero=True)
    if bounded:
        G = np.zeros((2 * m, n + m)); g = np.zeros(2 * m)
        bnd = np.array([1.5, 0.1, 0.1, 0.1])
        for i in range(m):
            G[i, n + i] = 1.0; g[i] = uref[i] + bnd[i]
            G[m + i, n + i] = -1.0; g[m + i] = -(uref[i] - bnd[i])
        for half in (0, 1):   # two blocks of 4 rows (the tile's dedicated lanes take dimension <= 4 per block)
            bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G[half * m:(half + 1) * m], g[half * m:(half + 1) * m])
    return bt


print(f"# tools/tile_variants.py: {batch} problems, N = {N}, (12, 4) tile plan, fp64; median of 15, host clock, ms")
print(f"{'dynamics':10s} {'cost':9s} {'constraints':12s} {'merit+deriv':>12s} {'backward':>10s} {'solve':>10s} {'sweeps':>7s} {'merit launches':>15s} {'converged':>10s}")
for model in (False, True):
    for dense in (False, True):
        for bounded in (False, True):
            bt = build(model, dense, bounded)
            bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
            t_merit = med(lambda: bt.merit(1.0, True))
            t_bwd = med(lambda: (bt.backward(), bt.synchronize()))
            res = None

            def solve():
                global res
                bt.set_input_guess((HOVER if model else np.zeros(m))[None, None], k_stride_zero=True, batch_stride_zero=True)
                if bounded:
                    bt.reset_duals(1.0)
                res = bt.ilqr_solve(iterations_max=40, tol_stationarity=1e-3)
            t_solve = med(solve, reps=5)
            print(f"{'quadrotor' if model else 'data':10s} {'dense' if dense else 'diagonal':9s} {'|u| bounds' if bounded else '-':12s} "
                  f"{t_merit:12.3f} {t_bwd:10.3f} {t_solve:10.2f} {res['sweeps']:7d} {res['merit_launches']:15d} {int((res['status'] == 0).sum()):10d}", flush=True)
            bt.close()
