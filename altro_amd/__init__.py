"""altro_amd -- MI355X-native ALTRO iLQR inner loop.

The product is the C-ABI shared library (include/altro_hip/altro_hip.h, built from altro_amd/csrc by
`python -m altro_amd.build`).  This module is only the ctypes binding a Python caller would write:
it holds no arithmetic and no CPU fallback -- if the library or a HIP device is missing, calls fail.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaltro_hip.so")

F64, F32 = 0, 1
PLAN_AUTO, PLAN_GENERIC, PLAN_MFMA16, PLAN_LANE, PLAN_MFMA32 = 0, 1, 2, 3, 4
STORE_QBLOCKS = 0x1
LANE_FUSED = 0x4   # plan LANE: FMA-fused TVLQR kernels (not bit-identical to the CPU path, ~20 % faster at small batch)
F32_PURE = 0x2   # ALTRO_HIP_F32 on plan MFMA16: backward sweep in pure fp32 (v_mfma_f32_16x16x4_f32)
GENERIC_MATRIX_CORES = 0x8   # plan GENERIC, fp64: the backward sweep's products on the matrix cores (equal to rounding, not bit for bit)
TVLQR_SUCCESS = -1

# every symbol include/altro_hip/altro_hip.h declares (tests check the .so exports all of them)
C_ABI_SYMBOLS = [
    "altro_hip_version", "altro_hip_last_error", "altro_hip_device_count", "altro_hip_device_info", "altro_hip_device_pci_bus_id",
    "altro_hip_batch_create", "altro_hip_batch_create_dims", "altro_hip_batch_destroy", "altro_hip_batch_plan",
    "altro_hip_batch_device_bytes", "altro_hip_set_dynamics", "altro_hip_set_cost",
    "altro_hip_set_initial_state", "altro_hip_set_host_batch", "altro_hip_backward", "altro_hip_forward_ltv", "altro_hip_sweep",
    "altro_hip_synchronize", "altro_hip_get_K", "altro_hip_get_d", "altro_hip_get_P",
    "altro_hip_get_p", "altro_hip_get_x", "altro_hip_get_u", "altro_hip_get_y",
    "altro_hip_get_delta_V", "altro_hip_get_status", "altro_hip_get_qblocks",
    "altro_hip_stats_reduce", "altro_hip_comm_unique_id", "altro_hip_comm_create", "altro_hip_comm_create_all",
    "altro_hip_comm_destroy", "altro_hip_comm_rank", "altro_hip_comm_world", "altro_hip_comm_device", "altro_hip_stats_allreduce", "altro_hip_stats_allreduce_multi",
    "altro_hip_profile_enable", "altro_hip_profile_reset",
    "altro_hip_profile_get", "altro_hip_profile_get_range", "altro_hip_profile_dropped", "altro_hip_algorithmic_bytes",
    "altro_hip_set_model", "altro_hip_set_model_source", "altro_hip_model_row_layout", "altro_hip_set_tracking_cost", "altro_hip_set_quadratic_cost",
    "altro_hip_set_input_guess", "altro_hip_set_state_guess",
    "altro_hip_open_loop_rollout", "altro_hip_accept", "altro_hip_expand", "altro_hip_merit",
    "altro_hip_stationarity", "altro_hip_get_nominal", "altro_hip_get_expansion",
    "altro_hip_default_solve_options", "altro_hip_set_forms", "altro_hip_get_forms", "altro_hip_ilqr_solve", "altro_hip_last_solve_counts",
    "altro_hip_ilqr_solve_async", "altro_hip_ilqr_poll", "altro_hip_ilqr_wait",
    "altro_hip_linesearch_host",
    "altro_hip_add_linear_constraint", "altro_hip_add_user_constraint", "altro_hip_clear_constraints", "altro_hip_reset_duals",
    "altro_hip_get_duals", "altro_hip_feasibility",
    "altro_hip_shift_trajectory", "altro_hip_update_linear_costs", "altro_hip_get_knot",
    "altro_hip_set_pointer_mode",
]
CONE_EQUALITY, CONE_IDENTITY, CONE_INEQUALITY, CONE_SOC = 0, 1, 2, 3   # ConstraintType, typedefs.hpp:29-34

# altro_hip_solve_options::forms / altro_hip_set_forms (ALTRO_HIP_FORM_*): how a solve is executed.  The C library reads no environment
# for these any more (version 300); THIS binding still translates the variables the test-suite and the tools set (forms_from_env), so
# that one process can alternate forms between calls.
FORM_NO_SPECULATION, FORM_NO_RUNAHEAD, FORM_NO_MERIT2, FORM_MERIT_LDS, FORM_MERIT_DPP_ALWAYS = 0x1, 0x2, 0x4, 0x8, 0x10
FORM_EXPAND_LDS, FORM_ALROWS_LDS, FORM_ROLLOUT_ROUNDS, FORM_SEQUENCED, FORM_MERIT_ONE_LAUNCH = 0x20, 0x40, 0x80, 0x100, 0x200
FORM_LANE_QUAD_OFF, FORM_LANE_QUAD_ON, FORM_GENERIC_LATE_Q_OFF, FORM_GENERIC_LATE_Q_ON, FORM_FUSED_CLOCK = 0x400, 0x800, 0x1000, 0x2000, 0x4000
FORM_AFFINE_EXACT = 0x8000
FORM_NO_COMPACTION = 0x10000
FORM_GENERIC_MERIT_LDS = 0x20000


def forms_from_env():
    """The ALTRO_HIP_* comparison switches of the environment as ALTRO_HIP_FORM_* bits (what the library itself read up to version 200)."""
    e = os.environ
    off = lambda name: e.get(name) is not None and e[name].strip() not in ("",) and _atoi(e[name]) == 0
    f = 0
    if "ALTRO_HIP_NO_SPECULATION" in e: f |= FORM_NO_SPECULATION
    if "ALTRO_HIP_NO_RUNAHEAD" in e and not off("ALTRO_HIP_NO_RUNAHEAD"): f |= FORM_NO_RUNAHEAD
    if off("ALTRO_HIP_MERIT2"): f |= FORM_NO_MERIT2
    if off("ALTRO_HIP_MERIT_DPP"): f |= FORM_MERIT_LDS
    elif e.get("ALTRO_HIP_MERIT_DPP") is not None and _atoi(e["ALTRO_HIP_MERIT_DPP"]) == 2: f |= FORM_MERIT_DPP_ALWAYS
    if off("ALTRO_HIP_EXPAND_DPP"): f |= FORM_EXPAND_LDS
    if off("ALTRO_HIP_ALROWS_DPP"): f |= FORM_ALROWS_LDS
    if off("ALTRO_HIP_AFFINE"): f |= FORM_ROLLOUT_ROUNDS
    if off("ALTRO_HIP_FUSED") or "ALTRO_HIP_NO_FUSED" in e: f |= FORM_SEQUENCED
    if off("ALTRO_HIP_MERIT_SPLIT"): f |= FORM_MERIT_ONE_LAUNCH
    if "ALTRO_HIP_LANE_QUAD" in e: f |= FORM_LANE_QUAD_OFF if _atoi(e["ALTRO_HIP_LANE_QUAD"]) == 0 else FORM_LANE_QUAD_ON
    if "ALTRO_HIP_GENERIC_LATE_Q" in e: f |= FORM_GENERIC_LATE_Q_OFF if _atoi(e["ALTRO_HIP_GENERIC_LATE_Q"]) == 0 else FORM_GENERIC_LATE_Q_ON
    if "ALTRO_HIP_FUSED_CLOCK" in e: f |= FORM_FUSED_CLOCK
    if "ALTRO_HIP_AFFINE_EXACT" in e: f |= FORM_AFFINE_EXACT
    if "ALTRO_HIP_NO_COMPACTION" in e: f |= FORM_NO_COMPACTION
    return f


def _atoi(text):
    try:
        return int(text.strip() or 0)
    except ValueError:
        return 0

MODEL_LINEAR, MODEL_DOUBLE_INTEGRATOR, MODEL_PENDULUM, MODEL_BICYCLE, MODEL_USER, MODEL_QUADROTOR, MODEL_QUADROTOR13 = 0, 1, 2, 3, 4, 5, 6
MERIT_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


class SolveOptions(C.Structure):
    _fields_ = [("iterations_max", C.c_int), ("tol_stationarity", C.c_double),
                ("tol_primal_feasibility", C.c_double), ("tol_meritfun_gradient", C.c_double),
                ("use_backtracking_linesearch", C.c_int), ("penalty_initial", C.c_double),
                ("penalty_scaling", C.c_double), ("penalty_max", C.c_double), ("reg_initial", C.c_double),
                ("reg_retry_max", C.c_int), ("reg_scale", C.c_double), ("reg_min", C.c_double),
                ("reg_max", C.c_double), ("stop_when_running_at_most", C.c_int),
                ("forms", C.c_uint), ("fused_sweeps", C.c_int), ("decision_margin", C.c_double)]   # version 300


class SolveResult(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("stationarity", C.c_double),
                ("final_alpha", C.c_double), ("final_phi", C.c_double), ("primal_feasibility", C.c_double),
                ("penalty", C.c_double), ("dual_updates", C.c_int), ("reg_retries", C.c_int)]


class PollRecord(C.Structure):
    """altro_hip_poll_record: one problem's results + first input, published by the solve kernel when the problem stops."""
    _fields_ = [("result", SolveResult), ("u0", C.c_double * 4), ("done", C.c_int), ("reserved", C.c_int)]


class AltroHipError(RuntimeError):
    pass


class Stats(C.Structure):
    """altro_hip_stats: what SolverImpl::Solve reports, reduced over a batch (and over GPUs)."""
    _fields_ = [("problems", C.c_int64), ("cholesky_failures", C.c_int64), ("converged", C.c_int64),
                ("iterations", C.c_int64), ("sum_cost", C.c_double),
                ("sum_delta_V0", C.c_double), ("sum_delta_V1", C.c_double),
                ("max_stationarity", C.c_double), ("max_feasibility", C.c_double), ("max_abs_xN", C.c_double),
                ("non_finite", C.c_int64)]
    SUM_FIELDS = ("problems", "cholesky_failures", "converged", "iterations", "sum_cost", "sum_delta_V0", "sum_delta_V1",
                  "non_finite")
    MAX_FIELDS = ("max_stationarity", "max_feasibility", "max_abs_xN")

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


COMM_ID_BYTES = 128


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64 with the same SONAME (libamdhip64.so.7) as /opt/rocm's.  Whichever
    copy a process loads first serves every later request for that SONAME -- but torch asks for it by file name, so a
    process that loaded the system copy through libaltro_hip.so and imports torch afterwards ends up with TWO HIP
    runtimes, and torch may then fail to find the GPU.  When torch is installed (it need not be imported), load its
    copy first so that this library and a later `import torch` share one runtime.  ALTRO_HIP_SYSTEM_RUNTIME=1 opts out."""
    import sys
    if "torch" in sys.modules or os.environ.get("ALTRO_HIP_SYSTEM_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass


_lib = None


def lib():
    """Loads the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AltroHipError("%s is missing: run `python -m altro_amd.build` (hipcc, gfx950). "
                                "There is no CPU fallback." % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.altro_hip_last_error.restype = C.c_char_p
        L.altro_hip_device_info.argtypes = [i, C.c_char_p, i, C.POINTER(i), C.POINTER(i)]
        L.altro_hip_device_pci_bus_id.argtypes = [i, C.c_char_p, i]
        L.altro_hip_batch_create.argtypes = [C.POINTER(vp), i, i, i, i, i, i, C.c_uint, i, vp]
        L.altro_hip_batch_create_dims.argtypes = [C.POINTER(vp), i, vp, vp, i, i, C.c_uint, i, vp]
        L.altro_hip_batch_destroy.argtypes = [vp]
        L.altro_hip_batch_destroy.restype = None
        L.altro_hip_batch_plan.argtypes = [vp]
        L.altro_hip_batch_device_bytes.argtypes = [vp]
        L.altro_hip_batch_device_bytes.restype = C.c_size_t
        L.altro_hip_set_dynamics.argtypes = [vp, vp, vp, vp, i, i]
        L.altro_hip_set_cost.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i]
        L.altro_hip_set_initial_state.argtypes = [vp, vp, i]
        L.altro_hip_set_host_batch.argtypes = [vp, i]
        L.altro_hip_backward.argtypes = [vp, d]
        L.altro_hip_forward_ltv.argtypes = [vp]
        L.altro_hip_sweep.argtypes = [vp, d]
        L.altro_hip_synchronize.argtypes = [vp]
        for g in ("K", "d", "P", "p", "x", "u", "y", "delta_V", "status", "qblocks"):
            getattr(L, "altro_hip_get_" + g).argtypes = [vp, vp]
        L.altro_hip_stats_reduce.argtypes = [vp, C.POINTER(Stats)]
        L.altro_hip_comm_unique_id.argtypes = [vp]
        L.altro_hip_comm_create.argtypes = [C.POINTER(vp), i, i, i, vp]
        L.altro_hip_comm_create_all.argtypes = [C.POINTER(vp), i, C.POINTER(i)]
        L.altro_hip_comm_destroy.argtypes = [vp]
        L.altro_hip_comm_destroy.restype = None
        for fn in ("rank", "world", "device"):
            getattr(L, "altro_hip_comm_" + fn).argtypes = [vp]
        L.altro_hip_stats_allreduce.argtypes = [vp, vp, C.POINTER(Stats)]
        L.altro_hip_stats_allreduce_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), i, C.POINTER(Stats)]
        L.altro_hip_profile_enable.argtypes = [vp, i]
        L.altro_hip_profile_reset.argtypes = [vp]
        L.altro_hip_profile_get.argtypes = [vp, i, C.POINTER(i), C.POINTER(d), C.POINTER(C.c_char_p)]
        L.altro_hip_profile_get_range.argtypes = [vp, i, C.POINTER(d), C.POINTER(d)]
        L.altro_hip_profile_dropped.argtypes = [vp, i]
        L.altro_hip_algorithmic_bytes.argtypes = [vp, i]
        L.altro_hip_algorithmic_bytes.restype = d
        L.altro_hip_set_model.argtypes = [vp, i, C.c_float, i, d, d]
        L.altro_hip_set_model_source.argtypes = [vp, C.c_char_p, C.c_float]
        L.altro_hip_model_row_layout.argtypes = [vp]
        L.altro_hip_set_tracking_cost.argtypes = [vp, vp, vp, vp, vp, i, i]
        L.altro_hip_set_quadratic_cost.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i]
        L.altro_hip_set_input_guess.argtypes = [vp, vp, i, i]
        L.altro_hip_set_state_guess.argtypes = [vp, vp, i, i]
        for fn in ("open_loop_rollout", "accept", "expand"):
            getattr(L, "altro_hip_" + fn).argtypes = [vp]
        L.altro_hip_merit.argtypes = [vp, vp, i, i, vp, vp]
        L.altro_hip_stationarity.argtypes = [vp, vp]
        L.altro_hip_get_nominal.argtypes = [vp, vp, vp]
        L.altro_hip_get_expansion.argtypes = [vp, vp, vp, vp, vp]
        L.altro_hip_add_linear_constraint.argtypes = [vp, i, i, i, i, vp, vp, i]
        L.altro_hip_add_user_constraint.argtypes = [vp, i, i, i, i, i]
        L.altro_hip_clear_constraints.argtypes = [vp]
        L.altro_hip_reset_duals.argtypes = [vp, d]
        L.altro_hip_get_duals.argtypes = [vp, i, i, vp]
        L.altro_hip_feasibility.argtypes = [vp, vp]
        L.altro_hip_set_pointer_mode.argtypes = [vp, i]
        L.altro_hip_shift_trajectory.argtypes = [vp]
        L.altro_hip_update_linear_costs.argtypes = [vp, vp, vp, vp, i, i, i, i]
        L.altro_hip_get_knot.argtypes = [vp, i, vp, vp]
        L.altro_hip_default_solve_options.argtypes = [C.POINTER(SolveOptions)]
        L.altro_hip_default_solve_options.restype = None
        L.altro_hip_set_forms.argtypes = [vp, C.c_uint]
        L.altro_hip_get_forms.argtypes = [vp]
        L.altro_hip_get_forms.restype = C.c_uint
        L.altro_hip_ilqr_solve.argtypes = [vp, C.POINTER(SolveOptions), vp]
        L.altro_hip_last_solve_counts.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
        L.altro_hip_ilqr_solve_async.argtypes = [vp, C.POINTER(SolveOptions)]
        L.altro_hip_ilqr_poll.argtypes = [vp, C.POINTER(i), C.POINTER(C.POINTER(PollRecord))]
        L.altro_hip_ilqr_wait.argtypes = [vp, vp]
        L.altro_hip_linesearch_host.argtypes = [MERIT_FN, vp, d, d, d, i, i, d, d, C.POINTER(i), C.POINTER(i),
                                                C.POINTER(d), C.POINTER(d)]
        L.altro_hip_linesearch_host.restype = d
        L.altro_hip_selftest_mfma_f64.argtypes = [i]
        L.altro_hip_selftest_mfma_f64.restype = d
        L.altro_hip_selftest_mfma_f32_4b.argtypes = [i]
        L.altro_hip_selftest_mfma_f32_4b.restype = d
        L.altro_hip_selftest_sincos.argtypes = [i, vp, i, vp, vp]
        L.altro_hip_selftest_sincos.restype = i
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise AltroHipError("altro_hip error %d: %s" % (rc, lib().altro_hip_last_error().decode()))


def device_info(device=0):
    """-> (name, compute units, PCI bus id) of a HIP device."""
    L = lib()
    name, pci = C.create_string_buffer(256), C.create_string_buffer(64)
    cus, ws = C.c_int(), C.c_int()
    _check(L.altro_hip_device_info(int(device), name, 256, C.byref(cus), C.byref(ws)))
    _check(L.altro_hip_device_pci_bus_id(int(device), pci, 64))
    return name.value.decode(), cus.value, pci.value.decode()


def _in(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.c_void_p)


class Batch:
    """`batch` independent TVLQR / iLQR problems on one MI355X (mirror of the C handle)."""

    def __init__(self, N, n, m, batch, dtype=F64, plan=PLAN_AUTO, flags=0, device=0, stream=None):
        self.L = lib()
        self.N, self.n, self.m, self.batch = N, n, m, batch
        self.h = C.c_void_p()
        _check(self.L.altro_hip_batch_create(C.byref(self.h), N, n, m, batch, dtype, plan, flags,
                                             device, stream))

    @property
    def plan(self):
        """The plan the handle runs NOW (a small-shape handle created with PLAN_AUTO moves from the padded tile to LANE when a LANE-only
        device model is set on it)."""
        return self.L.altro_hip_batch_plan(self.h)

    @classmethod
    def with_dims(cls, nx, nu, batch, dtype=F64, flags=0, device=0, stream=None):
        """altro_hip_batch_create_dims: per-knot-point dimensions nx[0..N], nu[0..N-1] (plan GENERIC: the TVLQR sweeps and the iLQR
        loop -- set_dynamics / set_quadratic_cost / set_initial_state / set_input_guess / add_linear_constraint / ilqr_solve ...).
        Bulk arrays are flat per problem, [batch, sum_k block_k]; `get` / `get_nominal` return them like that."""
        self = cls.__new__(cls)
        self.L = lib()
        self.nx = np.ascontiguousarray(nx, dtype=np.int32); self.nu = np.ascontiguousarray(nu, dtype=np.int32)
        self.N, self.n, self.m, self.batch = len(self.nu), int(self.nx.max()), int(self.nu.max()), batch
        assert len(self.nx) == self.N + 1
        self.h = C.c_void_p()
        _check(self.L.altro_hip_batch_create_dims(C.byref(self.h), self.N, self.nx.ctypes.data_as(C.c_void_p),
                                                  self.nu.ctypes.data_as(C.c_void_p), batch, dtype, flags, device, stream))
        return self

    def _ragged_len(self, name):
        nx, nu, N = self.nx.astype(np.int64), self.nu.astype(np.int64), self.N
        return {"K": int((nu * nx[:N]).sum()), "d": int(nu.sum()), "P": int((nx * nx).sum()), "p": int(nx.sum()),
                "x": int(nx.sum()), "u": int(nu.sum()), "y": int(nx.sum())}[name]

    def set_pointer_mode(self, device_pointers):
        """device_pointers=True: the bulk arrays handed to the raw C entry points are device pointers (see
        altro_hip.h).  The numpy-based helpers of this class always pass host arrays; use `self.L` + `self.h` with
        your own device pointers (e.g. torch tensors' data_ptr()) while the mode is on."""
        _check(self.L.altro_hip_set_pointer_mode(self.h, int(bool(device_pointers))))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.altro_hip_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def device_bytes(self):
        return self.L.altro_hip_batch_device_bytes(self.h)

    def set_dynamics(self, A, B, f=None, k_stride_zero=False, batch_stride_zero=False):
        (a, pa), (b, pb), (c, pc) = _in(A), _in(B), _in(f)
        _check(self.L.altro_hip_set_dynamics(self.h, pa, pb, pc, int(k_stride_zero), int(batch_stride_zero)))

    def set_cost(self, Q, R, H, q, r, is_diag=False, k_stride_zero=False, batch_stride_zero=False):
        keep = [_in(v) for v in (Q, R, H, q, r)]
        _check(self.L.altro_hip_set_cost(self.h, *[k[1] for k in keep], int(is_diag),
                                         int(k_stride_zero), int(batch_stride_zero)))

    def set_host_batch(self, host_batch):
        _check(self.L.altro_hip_set_host_batch(self.h, int(host_batch)))

    def set_initial_state(self, x0, batch_stride_zero=False):
        a, pa = _in(x0)
        _check(self.L.altro_hip_set_initial_state(self.h, pa, int(batch_stride_zero)))

    def backward(self, reg=0.0):
        self._sync_forms()
        _check(self.L.altro_hip_backward(self.h, float(reg)))

    def forward_ltv(self):
        self._sync_forms()
        _check(self.L.altro_hip_forward_ltv(self.h))

    def sweep(self, reg=0.0):
        self._sync_forms()
        _check(self.L.altro_hip_sweep(self.h, float(reg)))

    def synchronize(self):
        _check(self.L.altro_hip_synchronize(self.h))

    def get(self, name):
        N, n, m, B = self.N, self.n, self.m, self.batch
        shape = {"K": (B, N, m * n), "d": (B, N, m), "P": (B, N + 1, n * n), "p": (B, N + 1, n),
                 "x": (B, N + 1, n), "u": (B, N, m), "y": (B, N + 1, n), "delta_V": (B, 2),
                 "qblocks": (B, N, n * n + m * m + m * n + n + m)}
        if name == "status":
            out = np.zeros(B, dtype=np.int32)
        elif getattr(self, "nx", None) is not None and name in ("K", "d", "P", "p", "x", "u", "y"):
            out = np.zeros((B, self._ragged_len(name)), dtype=np.float64)
        else:
            out = np.zeros(shape[name], dtype=np.float64)
        _check(getattr(self.L, "altro_hip_get_" + name)(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self, comm=None):
        """This handle's statistics (device-side reduction); with a Comm, the global ones (two RCCL all-reduces)."""
        s = Stats()
        if comm is None:
            _check(self.L.altro_hip_stats_reduce(self.h, C.byref(s)))
        else:
            _check(self.L.altro_hip_stats_allreduce(self.h, comm.c, C.byref(s)))
        return s

    def profile(self, enable=True):
        """True / 1: hipEvents + a wait per launch; 2: events only (read by profile_get), for use inside a timed region."""
        _check(self.L.altro_hip_profile_enable(self.h, int(enable)))
        _check(self.L.altro_hip_profile_reset(self.h))

    def profile_range(self, slot):
        lo, hi = C.c_double(), C.c_double()
        _check(self.L.altro_hip_profile_get_range(self.h, slot, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def profile_dropped(self, slot):
        return self.L.altro_hip_profile_dropped(self.h, slot)

    def profile_get(self, slot):
        n, ms, name = C.c_int(), C.c_double(), C.c_char_p()
        _check(self.L.altro_hip_profile_get(self.h, slot, C.byref(n), C.byref(ms), C.byref(name)))
        return n.value, ms.value, name.value.decode()

    def algorithmic_bytes(self, slot):
        return self.L.altro_hip_algorithmic_bytes(self.h, slot)

    # ---- the iLQR loop around the sweep (device models; plan LANE shapes) ----
    def set_model(self, model, timestep, frame=0, length=2.7, lr=1.5):
        _check(self.L.altro_hip_set_model(self.h, model, float(timestep), frame, length, lr))

    def set_model_source(self, source, timestep):
        """The caller's own continuous dynamics + Jacobian as HIP source (altro_hip_set_model_source), compiled at run time."""
        _check(self.L.altro_hip_set_model_source(self.h, source.encode(), float(timestep)))

    def set_tracking_cost(self, Qd, Rd, xref, uref, k_stride_zero=False, batch_stride_zero=False):
        keep = [_in(v) for v in (Qd, Rd, xref, uref)]
        _check(self.L.altro_hip_set_tracking_cost(self.h, *[k[1] for k in keep], int(k_stride_zero),
                                                  int(batch_stride_zero)))

    def set_quadratic_cost(self, Q, R, H, q, r, c=None, k_stride_zero=False, batch_stride_zero=False):
        """ALTROSolver::SetQuadraticCost for the device iLQR loop: Q [.][N+1][n*n], R [.][N][m*m], H [.][N][m*n] column-major
        blocks, q [.][N+1][n], r [.][N][m], c [.][N+1] (None: zero); k_stride_zero: {running, terminal} / one knot point."""
        keep = [_in(v) for v in (Q, R, H, q, r)]
        kc = _in(c) if c is not None else (None, None)
        _check(self.L.altro_hip_set_quadratic_cost(self.h, *[k[1] for k in keep], kc[1], int(k_stride_zero),
                                                   int(batch_stride_zero)))

    def set_input_guess(self, u, k_stride_zero=False, batch_stride_zero=False):
        a, pa = _in(u)
        _check(self.L.altro_hip_set_input_guess(self.h, pa, int(k_stride_zero), int(batch_stride_zero)))

    def set_state_guess(self, x, k_stride_zero=False, batch_stride_zero=False):
        """ALTROSolver::SetState: x [.][N+1][n] into the candidate states."""
        a, pa = _in(x)
        _check(self.L.altro_hip_set_state_guess(self.h, pa, int(k_stride_zero), int(batch_stride_zero)))

    def open_loop_rollout(self):
        self._sync_forms()
        _check(self.L.altro_hip_open_loop_rollout(self.h))

    def accept(self):
        self._sync_forms()
        _check(self.L.altro_hip_accept(self.h))

    def expand(self):
        self._sync_forms()
        _check(self.L.altro_hip_expand(self.h))

    def model_row_layout(self):
        """Whether the handle's device model runs the row-layout model kernels (plans GENERIC / MFMA32; altro_hip_model_row_layout)."""
        self._sync_forms()
        return bool(self.L.altro_hip_model_row_layout(self.h))

    def merit(self, alpha, derivative=True):
        self._sync_forms()
        phi = np.zeros(self.batch); dphi = np.zeros(self.batch)
        if np.isscalar(alpha):
            a = np.array([float(alpha)]); uniform = 1
        else:
            a = np.ascontiguousarray(alpha, dtype=np.float64); uniform = 0
        _check(self.L.altro_hip_merit(self.h, a.ctypes.data_as(C.c_void_p), uniform, int(derivative),
                                      phi.ctypes.data_as(C.c_void_p), dphi.ctypes.data_as(C.c_void_p)))
        return phi, (dphi if derivative else None)

    def stationarity(self):
        self._sync_forms()
        out = np.zeros(self.batch)
        _check(self.L.altro_hip_stationarity(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_nominal(self):
        if getattr(self, "nx", None) is not None:   # per-knot-point dimensions: packed [batch, sum nx], [batch, sum nu]
            x = np.zeros((self.batch, self._ragged_len("x"))); u = np.zeros((self.batch, self._ragged_len("u")))
        else:
            x = np.zeros((self.batch, self.N + 1, self.n)); u = np.zeros((self.batch, self.N, self.m))
        _check(self.L.altro_hip_get_nominal(self.h, x.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p)))
        return x, u

    def get_expansion(self):
        B, N, n, m = self.batch, self.N, self.n, self.m
        A = np.zeros((B, N, n * n)); Bm = np.zeros((B, N, n * m)); lx = np.zeros((B, N + 1, n)); lu = np.zeros((B, N, m))
        _check(self.L.altro_hip_get_expansion(self.h, *[v.ctypes.data_as(C.c_void_p) for v in (A, Bm, lx, lu)]))
        return A, Bm, lx, lu

    def shift_trajectory(self):
        _check(self.L.altro_hip_shift_trajectory(self.h))

    def update_linear_costs(self, q, r, c, k_first, k_last, k_stride_zero=False, batch_stride_zero=False):
        """q [nb, nk, n], r [nb, nk, m] or None, c [nb, nk] or None (nk = 1 / nb = 1 when the stride flags are set)."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (q, r, c)]
        ptr = [None if a is None else a.ctypes.data_as(C.c_void_p) for a in arrs]
        _check(self.L.altro_hip_update_linear_costs(self.h, ptr[0], ptr[1], ptr[2], int(k_first), int(k_last),
                                                    int(k_stride_zero), int(batch_stride_zero)))

    def get_knot(self, k, want_u=True):
        nk = self.n if getattr(self, "nx", None) is None else int(self.nx[k])
        mk = self.m if getattr(self, "nx", None) is None or k >= self.N else int(self.nu[k])
        x = np.zeros((self.batch, nk))
        u = np.zeros((self.batch, mk)) if (want_u and k < self.N) else None
        _check(self.L.altro_hip_get_knot(self.h, int(k), x.ctypes.data_as(C.c_void_p),
                                         u.ctypes.data_as(C.c_void_p) if u is not None else None))
        return x, u

    def add_linear_constraint(self, k_first, k_last, cone, G, g):
        """c = G [x;u] - g in `cone` at knot points k_first..k_last (inclusive).  G: (p, n+m) numpy (row-major
        here, sent column-major); g: (p,) shared by the batch or (batch, p) per problem.  Returns the block id."""
        G = np.asarray(G, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        Gc = np.ascontiguousarray(G.T)
        per_problem = int(g.ndim == 2)
        if per_problem:
            assert g.shape == (self.batch, G.shape[0])
        rc = self.L.altro_hip_add_linear_constraint(self.h, int(k_first), int(k_last), int(cone), int(G.shape[0]),
                                                    Gc.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p),
                                                    per_problem)
        if rc < 0:
            _check(rc)
        return rc

    def add_user_constraint(self, k_first, k_last, cone, p, cid):
        """Block `cid` of the constraints the run-time compiled source defines (altro_hip_add_user_constraint)."""
        rc = self.L.altro_hip_add_user_constraint(self.h, int(k_first), int(k_last), int(cone), int(p), int(cid))
        if rc < 0:
            _check(rc)
        return rc

    def clear_constraints(self):
        _check(self.L.altro_hip_clear_constraints(self.h))

    def reset_duals(self, penalty=1.0):
        _check(self.L.altro_hip_reset_duals(self.h, float(penalty)))

    def get_duals(self, k, slot, p):
        out = np.zeros((self.batch, p))
        _check(self.L.altro_hip_get_duals(self.h, int(k), int(slot), out.ctypes.data_as(C.c_void_p)))
        return out

    def feasibility(self):
        self._sync_forms()
        out = np.zeros(self.batch)
        _check(self.L.altro_hip_feasibility(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def _solve_options(self, iterations_max=200, tol_stationarity=1e-4, tol_meritfun_gradient=1e-8,
                       use_backtracking=False, tol_primal_feasibility=1e-4, penalty_initial=1.0, penalty_scaling=10.0,
                       penalty_max=1e8, reg_initial=0.0, reg_retry_max=0, reg_scale=10.0, reg_min=1e-6, reg_max=1e8,
                       stop_when_running_at_most=0, forms=0, fused_sweeps=None, decision_margin=None):
        self._sync_forms()
        o = SolveOptions()
        self.L.altro_hip_default_solve_options(C.byref(o))
        o.forms = int(forms)
        if fused_sweeps is None and "ALTRO_HIP_FUSED_SWEEPS" in os.environ:
            fused_sweeps = _atoi(os.environ["ALTRO_HIP_FUSED_SWEEPS"]) or 1
        o.fused_sweeps = int(fused_sweeps or 0)
        if decision_margin is not None:
            o.decision_margin = float(decision_margin)
        o.tol_primal_feasibility = tol_primal_feasibility
        o.penalty_initial, o.penalty_scaling, o.penalty_max = penalty_initial, penalty_scaling, penalty_max
        o.reg_initial, o.reg_retry_max, o.reg_scale = reg_initial, reg_retry_max, reg_scale
        o.reg_min, o.reg_max = reg_min, reg_max
        o.stop_when_running_at_most = int(stop_when_running_at_most)
        o.iterations_max, o.tol_stationarity = iterations_max, tol_stationarity
        o.tol_meritfun_gradient, o.use_backtracking_linesearch = tol_meritfun_gradient, int(use_backtracking)
        return o

    def set_forms(self, forms):
        """altro_hip_set_forms: ALTRO_HIP_FORM_* bits every later call on this handle runs with (OR-ed with the environment's, see forms_from_env)."""
        self._forms_user = int(forms)
        self._sync_forms()

    def _sync_forms(self):
        want = getattr(self, "_forms_user", 0) | forms_from_env()
        if want != getattr(self, "_forms_set", 0):
            _check(self.L.altro_hip_set_forms(self.h, want))
            self._forms_set = want

    @staticmethod
    def _results(rec, sweeps, merit_launches):
        return dict(status=rec["status"].copy(), iterations=rec["iterations"].copy(),
                    stationarity=rec["stationarity"].copy(), alpha=rec["final_alpha"].copy(), phi=rec["final_phi"].copy(),
                    feasibility=rec["primal_feasibility"].copy(), penalty=rec["penalty"].copy(),
                    dual_updates=rec["dual_updates"].copy(), reg_retries=rec["reg_retries"].copy(),
                    sweeps=sweeps, merit_launches=merit_launches)

    def ilqr_solve(self, **options):
        o = self._solve_options(**options)
        res = (SolveResult * self.batch)()
        _check(self.L.altro_hip_ilqr_solve(self.h, C.byref(o), res))
        sw, ml = C.c_int(), C.c_int()
        self.L.altro_hip_last_solve_counts(self.h, C.byref(sw), C.byref(ml))
        rec = np.frombuffer(res, dtype=np.dtype(SolveResult))     # one view instead of 9 Python loops over the batch
        return self._results(rec, sw.value, ml.value)

    def ilqr_solve_async(self, **options):
        """Start the solve and return at once (altro_hip_ilqr_solve_async); follow with poll() / wait()."""
        o = self._solve_options(**options)
        _check(self.L.altro_hip_ilqr_solve_async(self.h, C.byref(o)))

    def poll(self):
        """(records published so far, numpy view [batch] of the pinned altro_hip_poll_record array -- fields result.*, u0, done)."""
        n, recs = C.c_int(), C.POINTER(PollRecord)()
        _check(self.L.altro_hip_ilqr_poll(self.h, C.byref(n), C.byref(recs)))
        buf = (PollRecord * self.batch).from_address(C.addressof(recs.contents))
        return n.value, np.frombuffer(buf, dtype=np.dtype(PollRecord))

    def wait(self):
        res = (SolveResult * self.batch)()
        _check(self.L.altro_hip_ilqr_wait(self.h, res))
        sw, ml = C.c_int(), C.c_int()
        self.L.altro_hip_last_solve_counts(self.h, C.byref(sw), C.byref(ml))
        return self._results(np.frombuffer(res, dtype=np.dtype(SolveResult)), sw.value, ml.value)


class Comm:
    """One RCCL communicator rank (altro_hip_comm).  `unique_id()` on rank 0, hand the 128 bytes to every rank by any
    channel (bench.py: a torch.distributed broadcast), then Comm(device, rank, world, id) on every rank."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * COMM_ID_BYTES)()
        _check(lib().altro_hip_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def __init__(self, device, rank, world, uid):
        self.L = lib()
        self.c = C.c_void_p()
        assert len(uid) == COMM_ID_BYTES
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(self.L.altro_hip_comm_create(C.byref(self.c), int(device), int(rank), int(world), C.cast(buf, C.c_void_p)))

    @property
    def rank(self):
        return self.L.altro_hip_comm_rank(self.c)

    @property
    def world(self):
        """Number of ranks (= GPUs) the communicator was built over: what a bench line reports as n_gpus."""
        return self.L.altro_hip_comm_world(self.c)

    def close(self):
        if getattr(self, "c", None) is not None and self.c:
            self.L.altro_hip_comm_destroy(self.c)
            self.c = None

    __del__ = close


def linesearch_host(fn, alpha0, phi0, dphi0, try_cubic_first=False, use_backtracking=False, c1=1e-4, c2=0.9):
    """Drives the batched solver's line-search state machine on the host with a Python merit callback."""
    L = lib()
    evals = []

    def cb(a, phi, dphi, ctx):
        p, dp = fn(a)
        evals.append(a)
        phi[0] = p
        if dphi:
            dphi[0] = dp

    st, it, ph, dph = C.c_int(), C.c_int(), C.c_double(), C.c_double()
    alpha = L.altro_hip_linesearch_host(MERIT_FN(cb), None, alpha0, phi0, dphi0, int(try_cubic_first),
                                        int(use_backtracking), c1, c2, C.byref(st), C.byref(it),
                                        C.byref(ph), C.byref(dph))
    return dict(alpha=alpha, status=st.value, iters=it.value, phi=ph.value, dphi=dph.value, evals=evals)
