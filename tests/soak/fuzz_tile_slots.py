#!/usr/bin/env python3
"""Soak: random constraint tables on plan MFMA16's six-slot knot-point records (kernels/al_types.h: AL_TILE_MAXC) -- blocks of 1..48
rows in the zero / orthant cones (dense or bound-type Jacobians), second-order cones of 2..4 rows, random knot-point ranges (so tables
are uniform, piecewise uniform or ragged, with 0..6 slots at a knot point), shared or per-problem right-hand sides, diagonal or dense
cost -- as whole AL-iLQR solves against the oracle, EVERY problem of every batch, with rollout line-search rounds (the oracle's
evaluation) and with the default's affine rounds; and, per case, phi / phi' / the expansion's gradient / feasibility at the initial
rollout against plan GENERIC's lane-per-row kernels on the same problem.

    python tests/soak/fuzz_tile_slots.py [cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import altro_amd  # noqa: E402
from tests import problems  # noqa: E402
from tests.test_gpu_ilqr_generic import make_oracle  # noqa: E402

n, m = 12, 4
w = n + m
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SLOTS = 6


def random_blocks(N, batch, scale):
    """Blocks until some knot point is nearly full.  Bounds are placed relative to `scale` (the unconstrained solution's size) so
    that some rows bind and most problems stay feasible."""
    used = np.zeros(N + 1, dtype=int)
    blocks = []
    for _ in range(int(rng.integers(2, 7))):
        kind = rng.choice(["box", "dense", "soc", "eq"], p=[0.4, 0.3, 0.15, 0.15])
        k0 = int(rng.integers(0, N)); k1 = int(rng.integers(k0, N + 1)) if rng.random() < 0.7 else N - 1
        if rng.random() < 0.5:
            k0, k1 = 0, N - 1
        if rng.random() < 0.15:
            k0 = k1 = N
        k1 = max(k0, min(k1, N))
        if kind == "soc":
            p = int(rng.integers(2, 5)); slots = 1
            G = np.zeros((p, w))
            cols = rng.choice(np.arange(n, w) if k1 < N else np.arange(0, n), size=p - 1, replace=False)
            for r, c in enumerate(cols):
                G[r, c] = 1.0
            g = np.zeros(p); g[p - 1] = -float(rng.uniform(0.6, 2.0)) * scale
            cone = altro_amd.CONE_SOC
        elif kind == "eq":
            p = int(rng.integers(1, 3)); slots = 1
            G = np.zeros((p, w))
            for r in range(p):
                G[r, int(rng.integers(n, w)) if k1 < N else int(rng.integers(0, n))] = 1.0
            g = rng.normal(size=p) * 0.05
            cone = altro_amd.CONE_EQUALITY
        elif kind == "box":
            nv = int(rng.integers(1, 13)); vs = rng.choice(w if k1 < N else n, size=min(nv, w if k1 < N else n), replace=False)
            p = 2 * len(vs); slots = (p + 7) // 8
            G = np.zeros((p, w))
            for r, c in enumerate(vs):
                G[r, c] = 1.0; G[len(vs) + r, c] = -1.0
            g = np.full(p, float(rng.uniform(0.5, 2.5)) * scale)
            cone = altro_amd.CONE_INEQUALITY
        else:
            p = int(rng.integers(1, 20)); slots = (p + 7) // 8
            G = rng.normal(size=(p, w)) * (rng.random((p, w)) < 0.4)
            if k1 >= N:
                G[:, n:] = 0.0
            G[np.abs(G).sum(axis=1) == 0, 0] = 1.0
            g = np.full(p, float(rng.uniform(1.0, 3.0)) * scale * max(1.0, np.abs(G).sum(axis=1).max()))
            cone = altro_amd.CONE_INEQUALITY
        if (used[k0:k1 + 1] + slots).max() > SLOTS:
            continue
        used[k0:k1 + 1] += slots
        per_problem = cone != altro_amd.CONE_SOC and rng.random() < 0.25
        if per_problem:
            g = g[None, :] * (1.0 + 0.1 * rng.random((batch, p)))
        blocks.append((k0, k1, cone, G, g))
    return blocks, int(used.max())


def build(p, N, batch, blocks, plan, dense, forms):
    bt = altro_amd.Batch(N, n, m, batch, plan=plan)
    bt.set_forms(forms)
    bt.set_dynamics(p["A"], p["B"], p["f"])
    if dense:
        bt.set_quadratic_cost(p["Q"], p["R"], p["H"], p["q"], p["r"], p["c"])
    else:
        bt.set_tracking_cost(p["Qd"], p["Rd"], p["xref"], p["uref"])
    bt.set_initial_state(p["x0"]); bt.set_input_guess(p["u0"])
    for (k0, k1, cone, G, g) in blocks:
        bt.add_linear_constraint(k0, k1, cone, G, g)
    return bt


tot = {"affine": [0, 0, 0.0], "rollout": [0, 0, 0.0]}
kern_worst = 0.0
slot_hist = np.zeros(SLOTS + 1, dtype=int)
bad_cases = 0
for it in range(cases):
    N = int(rng.integers(6, 60)); batch = int(rng.integers(3, 14)); dense = bool(rng.random() < 0.4)
    p = problems.ilqr12x4_problem(batch, N, True)
    p["x0"] = p["x0"] * float(rng.uniform(0.5, 1.5))
    if dense:
        p.update(problems.quadratic_cost(batch, N, n, m))
    bt = build(p, N, batch, [], altro_amd.PLAN_AUTO, dense, 0)
    bt.ilqr_solve(iterations_max=40)
    scale = float(np.abs(bt.get_nominal()[0]).max())
    bt.close()
    blocks, most = random_blocks(N, batch, scale)
    if not blocks:
        continue
    slot_hist[most] += 1
    # kernels against plan GENERIC at the initial rollout
    outs = []
    for plan in (altro_amd.PLAN_AUTO, altro_amd.PLAN_GENERIC):
        bt = build(p, N, batch, blocks, plan, dense, 0)
        bt.open_loop_rollout(); bt.accept(); bt.expand(); bt.backward()
        phi, dphi = bt.merit(np.linspace(0.1, 1.0, batch))
        _, _, lx, lu = bt.get_expansion()
        outs.append(np.concatenate([phi, dphi, lx.ravel(), lu.ravel(), bt.feasibility()]))
        bt.close()
    kd = float(np.abs(outs[0] - outs[1]).max() / max(1.0, np.abs(outs[1]).max()))
    kern_worst = max(kern_worst, kd)
    # whole solves against the oracle
    ref = []
    for b in range(batch):
        s = make_oracle(p, b, N, n, m, dense)
        for (k0, k1, cone, G, g) in blocks:
            for k in range(k0, k1 + 1):
                s.add_linear_constraint(k, cone, G, g[b] if g.ndim == 2 else g)
        s.L.oracle_ilqr_initialize(s.h)
        for k in range(N):
            s.L.oracle_ilqr_set_input(s.h, k, np.ascontiguousarray(p["u0"][b, k]))
        s.set_penalty(1.0, 10.0)
        s.L.oracle_ilqr_set_options(s.h, 60, 1e-4, 1e-4, 1e-8, 0)
        status, iters, _ = s.solve()
        ref.append((status, iters, s.get("x").reshape(N + 1, n).copy()))
    line = []
    for name, forms in (("affine", 0), ("rollout", altro_amd.FORM_ROLLOUT_ROUNDS)):
        bt = build(p, N, batch, blocks, altro_amd.PLAN_AUTO, dense, forms)
        assert bt.plan == altro_amd.PLAN_MFMA16
        res = bt.ilqr_solve(iterations_max=60, penalty_initial=1.0, penalty_scaling=10.0)
        x, _ = bt.get_nominal()
        off, worst = 0, 0.0
        for b in range(batch):
            st, its, xr = ref[b]
            if res["status"][b] != st or res["iterations"][b] != its:
                off += 1
            elif st == 0:
                worst = max(worst, float(np.abs(x[b] - xr).max()))
        tot[name][0] += off; tot[name][1] += batch; tot[name][2] = max(tot[name][2], worst)
        line.append("%s %d off, %.1e" % (name, off, worst))
        bt.close()
    conv = sum(1 for r in ref if r[0] == 0)
    print("case %2d: N %2d batch %2d %s, %d blocks, most slots %d, oracle converged %2d; kernels vs GENERIC %.1e; %s"
          % (it, N, batch, "dense" if dense else "diag ", len(blocks), most, conv, kd, "; ".join(line)), flush=True)
print("slots at the fullest knot point, cases per count:", dict((i, int(c)) for i, c in enumerate(slot_hist) if c))
print("kernels against plan GENERIC: worst relative difference %.2e" % kern_worst)
for name in ("affine", "rollout"):
    print("%s rounds: %d of %d problems end with another status / iteration count than the oracle; converged rest within %.2e"
          % (name, tot[name][0], tot[name][1], tot[name][2]))
