"""ctypes door into the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (altro_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "liblinesearch_ref.so")

dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
MERIT_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def build(force=False):
    """Compile oracle/_build/liboracle.so (and oracle/_ref when /root/reference exists)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = force or not os.path.exists(_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"],
                              stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/linesearch") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"], stdout=subprocess.DEVNULL)


def timing_lib():
    """The same C sources built `-O3 -march=native` (still `-ffp-contract=off`: same algorithm, same results) for
    bench.py's cpu_baseline leg, so that the CPU side is timed at its best.  Host-specific, hence built on the
    machine that runs it and named after its CPU; falls back to the portable build when that fails."""
    import hashlib
    try:
        cpu = "".join(l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags")))[:4096]
    except OSError:
        cpu = "unknown"
    tag = hashlib.sha1(cpu.encode()).hexdigest()[:10]
    path = os.path.join(_HERE, "_build", "liboracle_native_%s.so" % tag)
    srcs = [os.path.join(_HERE, f) for f in ("tvlqr_oracle.c",)]
    try:
        if not os.path.exists(path) or any(os.path.getmtime(x) > os.path.getmtime(path) for x in srcs):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-std=c11", "-shared", "-o", path]
                                  + srcs + ["-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.oracle_backward_batch.argtypes = [C.c_int] * 4 + [C.c_void_p] * 8 + [C.c_double, C.c_int] + [C.c_void_p] * 6
        L.oracle_backward_batch.restype = None
        L.oracle_forward_batch.argtypes = [C.c_int] * 4 + [C.c_void_p] * 11
        L.oracle_forward_batch.restype = None
        return L, "gcc -O3 -march=native"
    except (OSError, subprocess.CalledProcessError):
        return lib(), "gcc -O2"


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.oracle_backward_batch.argtypes = [C.c_int] * 4 + [C.c_void_p] * 8 + [C.c_double, C.c_int] + [C.c_void_p] * 6
        L.oracle_backward_batch.restype = None
        L.oracle_forward_batch.argtypes = [C.c_int] * 4 + [C.c_void_p] * 11
        L.oracle_forward_batch.restype = None
        L.oracle_ws_create.restype = C.c_void_p
        L.oracle_ws_create.argtypes = [C.c_int] * 3
        L.oracle_ws_destroy.argtypes = [C.c_void_p]
        L.oracle_backward_flat.argtypes = [C.c_void_p] + [C.c_void_p] * 8 + [C.c_double, C.c_int] + [C.c_void_p] * 6
        L.oracle_backward_flat.restype = C.c_int
        L.oracle_tvlqr_TotalMemSize.argtypes = [ip, ip, C.c_int, C.c_bool]
        L.oracle_tvlqr_TotalMemSize.restype = C.c_int
        L.oracle_ilqr_create.restype = C.c_void_p
        L.oracle_ilqr_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        for name in ("destroy", "initialize", "open_loop_rollout", "copy_trajectory",
                     "calc_cost_gradient", "calc_dynamics_expansions", "calc_expansions",
                     "linear_rollout"):
            getattr(L, "oracle_ilqr_" + name).argtypes = [C.c_void_p]
            getattr(L, "oracle_ilqr_" + name).restype = None
        L.oracle_ilqr_calc_cost.argtypes = [C.c_void_p]
        L.oracle_ilqr_calc_cost.restype = C.c_double
        L.oracle_ilqr_stationarity.argtypes = [C.c_void_p]
        L.oracle_ilqr_stationarity.restype = C.c_double
        L.oracle_ilqr_backward_pass.argtypes = [C.c_void_p]
        L.oracle_ilqr_backward_pass.restype = C.c_int
        L.oracle_ilqr_set_options.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]
        L.oracle_ilqr_set_bicycle.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.oracle_ilqr_set_linear_dynamics.argtypes = [C.c_void_p, dp, dp, C.c_void_p]
        L.oracle_ilqr_set_quadratic_cost.argtypes = [C.c_void_p, C.c_int, dp, C.c_void_p, C.c_void_p, dp, C.c_void_p, C.c_double]
        L.oracle_ilqr_set_diagonal_cost.argtypes = [C.c_void_p, C.c_int, dp, C.c_void_p, dp, C.c_void_p, C.c_double]
        L.oracle_ilqr_set_lqr_cost.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp]
        L.oracle_ilqr_set_initial_state.argtypes = [C.c_void_p, dp]
        L.oracle_ilqr_set_state.argtypes = [C.c_void_p, C.c_int, dp]
        L.oracle_ilqr_set_input.argtypes = [C.c_void_p, C.c_int, dp]
        L.oracle_ilqr_merit.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_ilqr_forward_pass.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_ilqr_forward_pass.restype = C.c_int
        L.oracle_ilqr_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_ilqr_solve.restype = C.c_int
        L.oracle_ilqr_add_linear_constraint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_ilqr_set_penalty_options.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.oracle_ilqr_feasibility.argtypes = [C.c_void_p]
        L.oracle_ilqr_set_linesearch_tolerances.argtypes = [C.c_void_p, C.c_double, C.c_double]
        for name in ("dual_update", "penalty_update", "refresh_expansions"):
            getattr(L, "oracle_ilqr_" + name).argtypes = [C.c_void_p]
            getattr(L, "oracle_ilqr_" + name).restype = None
        L.oracle_ilqr_feasibility.restype = C.c_double
        L.oracle_cone_projection.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_cone_jacobian.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_cone_hessian.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_ilqr_update_linear_costs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double]
        L.oracle_ilqr_shift_trajectory.argtypes = [C.c_void_p]
        L.oracle_ilqr_iterations.argtypes = [C.c_void_p]
        L.oracle_ilqr_merit_evals.argtypes = [C.c_void_p]
        L.oracle_ilqr_delta_V.argtypes = [C.c_void_p, C.c_int]
        L.oracle_ilqr_delta_V.restype = C.c_double
        for g in ("x", "u", "y", "x_cand", "u_cand", "y_cand", "K", "d", "P", "p", "A", "B",
                  "lx", "lu", "lxx", "luu", "lux"):
            getattr(L, "oracle_ilqr_get_" + g).argtypes = [C.c_void_p, dp]
        L.oracle_di_dynamics.argtypes = [dp, dp, dp, C.c_float, C.c_int]
        L.oracle_di_jacobian.argtypes = [dp, dp, dp, C.c_float, C.c_int]
        L.oracle_di_dynamics_hd.argtypes = [dp, dp, dp, C.c_double, C.c_int]
        L.oracle_di_jacobian_hd.argtypes = [dp, C.c_double, C.c_int]
        L.oracle_pendulum_dynamics.argtypes = [dp, dp, dp]
        L.oracle_pendulum_jacobian.argtypes = [dp, dp, dp]
        L.oracle_discrete_dynamics.argtypes = [C.c_void_p, dp, dp, dp, C.c_float]
        L.oracle_discrete_jacobian.argtypes = [C.c_void_p, dp, dp, dp, C.c_float]
        L.oracle_bicycle_dynamics.argtypes = [C.c_void_p, dp, dp, dp]
        L.oracle_bicycle_jacobian.argtypes = [C.c_void_p, dp, dp, dp]
        L.oracle_ls_defaults.argtypes = [C.c_void_p]
        L.oracle_ls_run.argtypes = [C.c_void_p, MERIT_FN, C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.oracle_ls_run.restype = C.c_double
        L.oracle_cubic_argmin_2pts.argtypes = [C.c_double] * 6 + [C.POINTER(C.c_double)]
        L.oracle_cubic_argmin_2pts.restype = C.c_int
        _lib = L
    return _lib


def ref_linesearch():
    """The REAL reference line search (oracle/_ref), or None when it was never built."""
    if not os.path.exists(_REF):
        return None
    R = C.CDLL(_REF)
    R.ref_ls_run.argtypes = [MERIT_FN, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int,
                             C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                             C.POINTER(C.c_double), C.POINTER(C.c_double)]
    R.ref_ls_run.restype = C.c_double
    return R


class Bicycle(C.Structure):
    _fields_ = [("frame", C.c_int), ("length", C.c_double), ("lr", C.c_double)]


class Model(C.Structure):
    _fields_ = [("kind", C.c_int), ("dim", C.c_int), ("bike", Bicycle)]


MODEL_DI, MODEL_PENDULUM, MODEL_BICYCLE, MODEL_QUADROTOR, MODEL_QUADROTOR13 = 0, 1, 2, 3, 4
DYN_LINEAR, DYN_MODEL = 0, 1
COST_QUADRATIC, COST_DIAGONAL = 0, 1
CONE_EQUALITY, CONE_IDENTITY, CONE_INEQUALITY, CONE_SOC = 0, 1, 2, 3


def make_model(kind, dim=0, frame=0, length=2.7, lr=1.5):
    return Model(kind, dim, Bicycle(frame, length, lr))


class LineSearch(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("alpha_max", C.c_double), ("beta_increase", C.c_double),
                ("beta_decrease", C.c_double), ("min_interval_size", C.c_double),
                ("c1", C.c_double), ("c2", C.c_double), ("try_cubic_first", C.c_int),
                ("use_backtracking", C.c_int), ("status", C.c_int), ("n_iters", C.c_int),
                ("sufficient_decrease", C.c_int), ("curvature", C.c_int), ("phi", C.c_double),
                ("dphi", C.c_double), ("phi0", C.c_double), ("dphi0", C.c_double),
                ("phi_lo", C.c_double), ("phi_hi", C.c_double), ("dphi_lo", C.c_double),
                ("dphi_hi", C.c_double)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def backward_batch(A, B, f, Q, R, H, q, r, reg=0.0, is_diag=False, L=None):
    """Arrays are [batch, k, ...] in the reference's column-major blocks, flattened:
    A [b,N,n*n], B [b,N,n*m], f [b,N,n], Q [b,N+1,n*n or n], R [b,N,m*m or m], H [b,N,m*n],
    q [b,N+1,n], r [b,N,m]. Returns dict(K,d,P,p,dV,status)."""
    L = L or lib()
    batch, N = A.shape[0], A.shape[1]
    n = f.shape[2]
    m = r.shape[2]
    args = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None
            for a in (A, B, f, Q, R, H, q, r)]
    K = np.zeros((batch, N, n * m)); d = np.zeros((batch, N, m))
    P = np.zeros((batch, N + 1, n * n)); p = np.zeros((batch, N + 1, n))
    dV = np.zeros((batch, 2)); status = np.zeros(batch, dtype=np.int32)
    L.oracle_backward_batch(N, n, m, batch, *[_p(a) for a in args], float(reg), int(is_diag),
                            _p(K), _p(d), _p(P), _p(p), _p(dV), _p(status))
    return dict(K=K, d=d, P=P, p=p, dV=dV, status=status)


def forward_batch(A, B, f, K, d, P, p, x0, want_y=True, L=None):
    L = L or lib()
    batch, N = A.shape[0], A.shape[1]
    n = f.shape[2]
    m = d.shape[2]
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (A, B, f, K, d, P, p, x0)]
    x = np.zeros((batch, N + 1, n)); u = np.zeros((batch, N, m))
    y = np.zeros((batch, N + 1, n)) if want_y else None
    L.oracle_forward_batch(N, n, m, batch, *[_p(a) for a in args], _p(x), _p(u), _p(y))
    return dict(x=x, u=u, y=y)


class ILQR:
    """Thin handle over oracle_ilqr_* (restatement of SolverImpl, unconstrained)."""

    def __init__(self, N, n, m, h, dyn_kind, model_kind=0, model_dim=0, cost_kind=COST_DIAGONAL):
        self.L = lib()
        self.N, self.n, self.m = N, n, m
        self.h = self.L.oracle_ilqr_create(N, n, m, float(h), dyn_kind, model_kind, model_dim, cost_kind)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_ilqr_destroy(self.h)
            self.h = None

    def get(self, name):
        N, n, m = self.N, self.n, self.m
        shape = {"x": (N + 1, n), "u": (N, m), "y": (N + 1, n), "x_cand": (N + 1, n),
                 "u_cand": (N, m), "y_cand": (N + 1, n), "K": (N, n * m), "d": (N, m),
                 "P": (N + 1, n * n), "p": (N + 1, n), "A": (N, n * n), "B": (N, n * m),
                 "lx": (N + 1, n), "lu": (N, m), "lxx": (N + 1, n * n), "luu": (N, m * m),
                 "lux": (N, n * m)}[name]
        out = np.zeros(shape)
        getattr(self.L, "oracle_ilqr_get_" + name)(self.h, out)
        return out

    def merit(self, alpha, deriv=True):
        phi, dphi = C.c_double(), C.c_double()
        self.L.oracle_ilqr_merit(self.h, float(alpha), C.byref(phi), C.byref(dphi) if deriv else None)
        return phi.value, (dphi.value if deriv else None)

    def forward_pass(self):
        a = C.c_double()
        err = self.L.oracle_ilqr_forward_pass(self.h, C.byref(a))
        return err, a.value

    def add_linear_constraint(self, k, cone, G, g):
        """c(x,u) = G [x;u] - g in `cone` (CONE_*); G is (p, n+m) row-major numpy."""
        G = np.asarray(G, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        Gc = np.ascontiguousarray(G.T)   # column-major p x (n+m)
        rc = self.L.oracle_ilqr_add_linear_constraint(self.h, int(k), int(cone), int(G.shape[0]), _p(Gc), _p(g))
        assert rc >= 0
        return rc

    def set_penalty(self, initial=1.0, scaling=10.0, pmax=1e8):
        self.L.oracle_ilqr_set_penalty_options(self.h, float(initial), float(scaling), float(pmax))

    def feasibility(self):
        return self.L.oracle_ilqr_feasibility(self.h)

    def solve(self, log_cap=256):
        log = np.zeros((log_cap, 8))   # alpha, phi0, phi, dphi0, stationarity, ls_iters, feasibility, rho
        status = self.L.oracle_ilqr_solve(self.h, _p(log), log_cap)
        it = self.L.oracle_ilqr_iterations(self.h)
        return status, it, log[:it]
