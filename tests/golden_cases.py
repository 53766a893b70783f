"""Problem definitions of the dense golden fixtures (tests/golden/dense_fixtures.npz): shared by the generator
(tests/golden/make_dense_fixtures.py, which runs the oracle) and by the tests that replay the fixtures -- the GPU one
(tests/test_gpu_golden.py) must not need the oracle, so nothing here imports it."""
import hashlib

import numpy as np

from tests import problems

P_KNOTS = {"c1": [0, 128, 256], "c4": [0, 256, 512]}


def checksum(pr):
    h = hashlib.sha256()
    for k in sorted(pr):
        if isinstance(pr[k], np.ndarray):
            h.update(k.encode()); h.update(np.ascontiguousarray(pr[k]).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def tvlqr_problem(name):
    """The TVLQR inputs of fixture `name` (shared with the tests)."""
    if name == "c1":
        return problems.c1_double_integrator(2, N=256)
    if name == "c4":      # configs[4]: fp32 data (the oracle runs in fp64 on the fp32-rounded inputs)
        pr = problems.random_ltv(2, 512, 12, 4)
        return {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in pr.items()}
    if name == "c2shape":
        return problems.random_ltv(3, 100, 2, 1)
    if name == "c3shape":
        return problems.random_ltv(3, 50, 4, 2)
    raise KeyError(name)


def solve_case(name):
    """Solve fixture `name`: model (altro_hip_model id + name), N, n, m, h, diagonal costs, goal, initial states, input
    guess, double-integrator dimension, backtracking line search flag, iterations_max."""
    f32 = np.float32
    if name in ("di_n10", "di_n50"):     # BASELINE.json configs[0]: double_integrator_test.cpp:69-85 at N = 10 and N = 50
        N = 10 if name == "di_n10" else 50
        return dict(model_name="double_integrator", model=1, N=N, n=4, m=2, h=f32(f32(5.0) / f32(N)), Qd=[1.0] * 4, Rd=[1e-2] * 2,
                    Qfd=[1.0] * 4, xf=[0.0] * 4, x0s=np.array([[1.0, 2.0, 0.0, 0.0]]), u0=[0.0, 0.0], dim=2, bt=0, itmax=3)
    if name == "pendulum":
        return dict(model_name="pendulum", model=2, N=50, n=2, m=1, h=f32(f32(3.0) / 50.0), Qd=[1e-2, 1e-2], Rd=[1e-3],
                    Qfd=[1.0, 1.0], xf=[np.pi, 0.0], x0s=np.array([[-0.4, 0.0], [0.0, 0.0], [0.3, 0.0]]), u0=[0.1], dim=0,
                    bt=0, itmax=30)
    if name == "bicycle":
        return dict(model_name="bicycle", model=3, N=30, n=4, m=2, h=f32(f32(3.0) / 30.0), Qd=[1e-2] * 4, Rd=[1e-3] * 2,
                    Qfd=[10.0] * 4, xf=[1, 2, np.pi / 2, 0.0], x0s=0.02 * (np.arange(12).reshape(3, 4) % 7 - 3) / 3.0,
                    u0=[0.5, 0.0], dim=0, bt=1, itmax=30)
    raise KeyError(name)


TVLQR = ("c1", "c4", "c2shape", "c3shape")
SOLVES = ("di_n10", "di_n50", "pendulum", "bicycle")
