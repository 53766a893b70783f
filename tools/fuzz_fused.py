#!/usr/bin/env python3
"""Soak: the fused solve kernel against the launch-sequenced loop, bit for bit, over random shapes -- model, horizon,
batch (all three workgroup sizes, ragged tails), line search, element type, constraints on / off, regularisation retry,
hand-over after a random number of fused sweeps, listed launches over the still-running problems after
every few sweeps (the straggler compaction, forced).    python tools/fuzz_fused.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altro_amd                               # noqa: E402
from tests import problems                     # noqa: E402
from tests.test_gpu_fused import _same, _solve  # noqa: E402

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)


def make_case():
    model = rng.choice(["pendulum", "bicycle", "di2", "di4", "di6"])
    N = int(rng.integers(1, 70))
    batch = int(rng.choice([1, 7, 33, 64, 65, 200, 777, 2048, 2049, 3000, 4097, 6000, 9000]))
    dtype = altro_amd.F32 if rng.random() < 0.25 else altro_amd.F64
    constrained = bool(rng.random() < 0.6)
    spread = float(rng.uniform(0.1, 1.5))
    seed = int(rng.integers(1, 1000))

    def make():
        if model == "pendulum":
            bt = altro_amd.Batch(N, 2, 1, batch, dtype=dtype)
            bt.set_model(altro_amd.MODEL_PENDULUM, np.float32(0.04))
            xf = np.array([np.pi, 0.0])
            bt.set_tracking_cost(np.array([[1e-2, 1e-2], [1.0, 1.0]]), np.array([[1e-3]]), np.stack([xf, xf]), np.zeros((1, 1)),
                                 k_stride_zero=True, batch_stride_zero=True)
            if constrained:
                G = np.zeros((2, 3)); G[0, 2] = 1.0; G[1, 2] = -1.0
                bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2, 3.0))
            x0 = np.zeros((batch, 2)); x0[:, 0] = spread * (problems.uniform01((batch,), seed) - 0.5)
            bt.set_initial_state(x0)
            bt.set_input_guess(np.array([[[0.1]]]), k_stride_zero=True, batch_stride_zero=True)
        elif model == "bicycle":
            n, m = 4, 2
            x_ref, u_ref = problems.bicycle_reference(N + 1)
            bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
            bt.set_model(altro_amd.MODEL_BICYCLE, np.float32(0.1))
            bt.set_tracking_cost(np.full((1, N + 1, n), 1e-2), np.full((1, N, m), 1e-3), x_ref[None, :N + 1], u_ref[None, :N],
                                 batch_stride_zero=True)
            if constrained:
                G = np.zeros((2, n + m)); G[0, 3] = 1.0; G[1, 3] = -1.0
                bt.add_linear_constraint(0, N, altro_amd.CONE_INEQUALITY, G, np.full(2, np.pi / 3))
            bt.set_initial_state(x_ref[0] + (problems.uniform01((batch, n), seed, 0) - 0.5) * spread)
            bt.set_input_guess(np.array([[[u_ref[0][0], 0.0]]]), k_stride_zero=True, batch_stride_zero=True)
        else:
            dim = {"di2": 1, "di4": 2, "di6": 3}[model]
            n, m = 2 * dim, dim
            bt = altro_amd.Batch(N, n, m, batch, dtype=dtype)
            bt.set_model(altro_amd.MODEL_DOUBLE_INTEGRATOR, np.float32(0.2))
            bt.set_tracking_cost(np.ones((2, n)), np.full((1, m), 1e-2), np.zeros((2, n)), np.zeros((1, m)), k_stride_zero=True,
                                 batch_stride_zero=True)
            if constrained:
                G = np.zeros((2 * m, n + m)); G[:m, n:] = np.eye(m); G[m:, n:] = -np.eye(m)
                bt.add_linear_constraint(0, N - 1, altro_amd.CONE_INEQUALITY, G, np.full(2 * m, 0.8))
            bt.set_initial_state(2.0 * spread * (problems.uniform01((batch, n), seed, 0) - 0.5))
            bt.set_input_guess(np.zeros((1, 1, m)), k_stride_zero=True, batch_stride_zero=True)
        return bt
    opts = dict(iterations_max=int(rng.integers(1, 30)), use_backtracking=bool(rng.random() < 0.5))
    if rng.random() < 0.3:
        opts.update(reg_retry_max=3, reg_min=1e-3, reg_scale=10.0)
    if constrained:
        opts.update(penalty_scaling=float(rng.choice([10.0, 100.0])))
    return "%s N=%d batch=%d %s %s" % (model, N, batch, "f32" if dtype == altro_amd.F32 else "f64",
                                      "constrained" if constrained else "free"), make, opts


for c in range(CASES):
    name, make, opts = make_case()
    seq = _solve(make, {"ALTRO_HIP_NO_FUSED": "1"}, **opts)
    fused = _solve(make, {"ALTRO_HIP_FUSED": "1"}, **opts)
    hand = _solve(make, {"ALTRO_HIP_FUSED": "1", "ALTRO_HIP_FUSED_SWEEPS": str(int(rng.integers(1, 6)))}, **opts)
    listed = _solve(make, {"ALTRO_HIP_FUSED": "1", "ALTRO_HIP_FUSED_SWEEPS": str(-int(rng.integers(1, 5)))}, **opts)   # straggler compaction, forced
    _same(seq, fused)
    _same(seq, hand)
    _same(seq, listed)
    for r in (seq, fused, hand, listed):
        r[5].close()
    st = np.asarray(fused[0]["status"])
    print("%-46s %-70s sweeps %3d  converged %d/%d" % (name, str(opts), int(fused[0]["sweeps"]), int((st == 0).sum()), len(st)))
print("ok: %d cases bit-identical" % CASES)
