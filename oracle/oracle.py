"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

ctypes wrapper over oracle/_build/liboracle.so (the CPU restatement of diffsol's BDF / SDIRK path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

MODEL_EXPONENTIAL_DECAY = 0
MODEL_EXPONENTIAL_DECAY_ALGEBRAIC = 1
MODEL_EXPONENTIAL_DECAY_ALGEBRAIC_BATCHED = 2
MODEL_ROBERTSON_ODE = 3
MODEL_ROBERTSON_DAE = 4
MODEL_DYDT_Y2 = 5
MODEL_GAUSSIAN_DECAY = 6
MODEL_HEAT1D = 7
MODEL_RLC = 8
MODEL_EXPONENTIAL_DECAY_ROOT = 9
MODEL_SPM = 10
MODEL_HEAT2D = 11   # n = size^2, DAE, p = [diffusion scale]
MODEL_FOODWEB = 12  # n = 2 size^2, DAE, p = [alpha, beta]

METHOD_BDF = 0
METHOD_TR_BDF2 = 1
METHOD_ESDIRK34 = 2

STAT_NAMES = [
    "number_of_linear_solver_setups", "number_of_steps", "number_of_error_test_failures",
    "number_of_nonlinear_solver_iterations", "number_of_nonlinear_solver_fails",
    "number_of_linear_solver_setups_from_checkpoint", "number_of_linear_solver_setups_from_first_convergence_fail",
    "number_of_linear_solver_setups_from_second_convergence_fail", "number_of_linear_solver_setups_from_error_test_fail",
    "number_of_linear_solver_setups_from_step_success", "number_of_calls", "number_of_jac_muls", "number_of_matrix_evals",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_long)


def build(force=False):
    """Compile the C++ restatement (g++, a few seconds)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))
    ):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_solver_create.restype = C.c_void_p
        L.orc_solver_create.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_solver_destroy.argtypes = [C.c_void_p]
        L.orc_nstates.argtypes = [C.c_void_p]
        L.orc_nbatch.argtypes = [C.c_void_p]
        L.orc_step.argtypes = [C.c_void_p]
        L.orc_set_stop_time.argtypes = [C.c_void_p, C.c_double]
        L.orc_interpolate.argtypes = [C.c_void_p, C.c_double, _dp]
        L.orc_interpolate_dy.argtypes = [C.c_void_p, C.c_double, _dp]
        L.orc_get_state.argtypes = [C.c_void_p, _dp, _dp, _ip, _dp, _dp]
        L.orc_bdf_get_diff.argtypes = [C.c_void_p, _dp]
        L.orc_root_info.argtypes = [C.c_void_p, _dp, _ip]
        L.orc_stats.argtypes = [C.c_void_p, _lp]
        L.orc_solve_to_points.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.orc_solve.restype = C.c_long
        L.orc_solve.argtypes = [C.c_void_p, C.c_double, _dp]
        L.orc_solve_ensemble_independent.restype = C.c_double
        L.orc_solve_ensemble_independent.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_int, C.c_double, C.c_double,
                                                     C.c_int, C.c_double, C.c_int, _dp, _lp]
        L.orc_compute_r.argtypes = [C.c_int, C.c_double, _dp]
        L.orc_lu_solve.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _ip]
        L.orc_lu_solve_fullpiv.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.orc_squared_norm.restype = C.c_double
        L.orc_squared_norm.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double]
        L.orc_convergence_trace.argtypes = [C.c_double, C.c_double, C.c_int, _dp, C.c_int, _ip, _dp]
        L.orc_model_rhs.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, _dp]
        L.orc_model_jac_mul.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, _dp, _dp]
        L.orc_model_root.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, _dp]
        L.orc_load_external_model.argtypes = [C.c_char_p]
        L.orc_load_external_model.restype = C.c_int
        L.orc_model_dims.argtypes = [C.c_int, C.c_int, _ip]
        L.orc_model_init.argtypes = [C.c_int, C.c_int, _dp, C.c_double, _dp]
        L.orc_model_mass_gemv.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double, _dp]
        L.orc_model_out.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, _dp]
        L.orc_model_out.restype = C.c_int
        L.orc_model_root.restype = C.c_int
        L.orc_det_fn.argtypes = [C.c_int, C.c_double]
        L.orc_det_fn.restype = C.c_double
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


class OracleError(RuntimeError):
    pass


class OracleSolver:
    """One (possibly batched, lock-step) solver instance of the CPU restatement."""

    def __init__(self, model, p, *, nbatch=1, model_size=0, rtol=1e-6, atol=(1e-6,), t0=0.0, h0=1.0, method=METHOD_BDF, sens=False, sens_rtol=None,
                 sens_atol=None, options=None):
        """sens=True: problem.bdf_sens() — forward sensitivities integrated alongside; sens_rtol / sens_atol put them into the error control
        (None: turn_off_sensitivities_error_control)."""
        L = lib()
        p_arr, p_ptr = _d(np.asarray(p, dtype=np.float64).reshape(-1))
        a_arr, a_ptr = _d(np.asarray(atol, dtype=np.float64).reshape(-1))
        for k, v in (options or {}).items():  # problem.ode_options.<k> = v (max_nonlinear_solver_failures, max_error_test_failures)
            L.orc_next_solver_option(k.encode(), C.c_double(float(v)))
        if sens:
            sa = np.zeros(0) if sens_atol is None else np.asarray(sens_atol, dtype=np.float64).reshape(-1)
            sa_arr, sa_ptr = _d(sa if sa.size else np.zeros(1))
            L.orc_solver_create_sens.restype = C.c_void_p
            self._h = L.orc_solver_create_sens(C.c_int(model), C.c_int(model_size), C.c_int(nbatch), p_ptr, C.c_int(p_arr.size), C.c_double(rtol), a_ptr,
                                               C.c_int(a_arr.size), C.c_double(t0), C.c_double(h0), C.c_int(method),
                                               C.c_double(0.0 if sens_rtol is None else sens_rtol), sa_ptr, C.c_int(sa.size))
        else:
            self._h = L.orc_solver_create(model, model_size, nbatch, p_ptr, p_arr.size, rtol, a_ptr, a_arr.size, t0, h0, method)
        if not self._h:
            raise OracleError(L.orc_last_error().decode())
        self.n = L.orc_nstates(self._h)
        self.nbatch = L.orc_nbatch(self._h)
        self.sens = bool(sens)

    def interpolate_sens(self, t=None):
        """OdeSolverMethod::interpolate_sens: [nparams, nbatch, n]; t=None: state.s at the current time."""
        L = lib()
        npar = L.orc_nparams(C.c_void_p(self._h))
        out = np.empty((npar, self.nbatch, self.n))
        r = L.orc_interpolate_sens(C.c_void_p(self._h), C.c_double(np.nan if t is None else t), out.ctypes.data_as(_dp))
        if r < 0:
            raise OracleError(f"oracle interpolate_sens failed with {r}")
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_solver_destroy(self._h)
            self._h = None

    def step(self):
        """Returns 0 InternalTimestep, 1 RootFound, 2 TstopReached; raises on solver error."""
        r = lib().orc_step(self._h)
        if r < 0:
            raise OracleError(f"oracle step failed with OdeErr {-r}")
        return r

    def set_stop_time(self, t):
        r = lib().orc_set_stop_time(self._h, t)
        if r < 0:
            raise OracleError(f"oracle set_stop_time failed with OdeErr {-r}")

    def state(self):
        t = C.c_double()
        h = C.c_double()
        order = C.c_int()
        y = np.empty((self.nbatch, self.n))
        dy = np.empty((self.nbatch, self.n))
        lib().orc_get_state(self._h, C.byref(t), C.byref(h), C.byref(order), y.ctypes.data_as(_dp), dy.ctypes.data_as(_dp))
        return dict(t=t.value, h=h.value, order=order.value, y=y, dy=dy)

    def diff(self):
        out = np.empty((self.nbatch, 8, self.n))
        if lib().orc_bdf_get_diff(self._h, out.ctypes.data_as(_dp)) != 0:
            raise OracleError("not a BDF solver")
        return out

    def interpolate(self, t):
        y = np.empty((self.nbatch, self.n))
        r = lib().orc_interpolate(self._h, t, y.ctypes.data_as(_dp))
        if r < 0:
            raise OracleError(f"oracle interpolate failed with OdeErr {-r}")
        return y

    def root_info(self):
        t = C.c_double()
        i = C.c_int()
        lib().orc_root_info(self._h, C.byref(t), C.byref(i))
        return t.value, i.value

    def interpolate_dy(self, t):
        dy = np.empty((self.nbatch, self.n))
        r = lib().orc_interpolate_dy(self._h, C.c_double(t), dy.ctypes.data_as(_dp))
        if r < 0:
            raise OracleError(f"oracle interpolate_dy failed with OdeErr {-r}")
        return dy

    def stats(self):
        out = (C.c_long * 13)()
        lib().orc_stats(self._h, out)
        return dict(zip(STAT_NAMES, [int(v) for v in out]))

    def solve_to_points(self, t_points):
        """The reference's test_ode_solver loop (use_tstop=False): step past each point, interpolate there."""
        tp, tp_ptr = _d(t_points)
        out = np.empty((tp.size, self.nbatch, self.n))
        r = lib().orc_solve_to_points(self._h, tp_ptr, tp.size, out.ctypes.data_as(_dp))
        if r < 0:
            raise OracleError(f"oracle solve_to_points failed with OdeErr {-r}")
        return out, r

    def solve(self, t_final):
        """OdeSolverMethod::solve: returns (final state.y [nbatch, n], number of output columns)."""
        y = np.empty((self.nbatch, self.n))
        r = lib().orc_solve(self._h, t_final, y.ctypes.data_as(_dp))
        if r < 0:
            raise OracleError(f"oracle solve failed with OdeErr {-r}")
        return y, int(r)


def solve_ensemble_independent(model, p, *, model_size=0, rtol=1e-6, atol=(1e-6,), t0=0.0, h0=1.0, method=METHOD_BDF, t_final=1.0,
                               nthreads=1, want_y=True):
    """CPU baseline: one independent IVP per parameter set (the reference's CPU usage pattern), over `nthreads` threads."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    nsys, np_ = p.shape
    a_arr, a_ptr = _d(np.asarray(atol, dtype=np.float64).reshape(-1))
    # nstates via a throw-away solver
    s = OracleSolver(model, p[0], model_size=model_size, rtol=rtol, atol=atol, t0=t0, h0=h0, method=method)
    n = s.n
    del s
    y = np.empty((nsys, n)) if want_y else None
    counters = (C.c_long * 4)()
    secs = lib().orc_solve_ensemble_independent(model, model_size, nsys, p.ctypes.data_as(_dp), np_, rtol, a_ptr, a_arr.size, t0, h0, method,
                                                t_final, nthreads, y.ctypes.data_as(_dp) if want_y else None, counters)
    return dict(seconds=secs, steps=int(counters[0]), newton_iterations=int(counters[1]), lu_setups=int(counters[2]),
                failed=int(counters[3]), y=y)


def solve_ensemble_independent_fast(model, p, *, model_size=0, rtol=1e-6, atol=(1e-6,), t0=0.0, h0=1.0, t_final=1.0, nthreads=1, want_y=True,
                                    want_stats=False):
    """solve_ensemble_independent on the stack-array build of the BDF (oracle_fast.hpp: same arithmetic, no per-operation allocation) — ODE
    models with identity mass, no roots, n in (3, 4).  `stats` [nsys, 5] per member when want_stats."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    nsys, np_ = p.shape
    a_arr, a_ptr = _d(np.asarray(atol, dtype=np.float64).reshape(-1))
    s = OracleSolver(model, p[0], model_size=model_size, rtol=rtol, atol=atol, t0=t0, h0=h0)
    n = s.n
    del s
    y = np.empty((nsys, n)) if want_y else None
    stats = np.zeros((nsys, 5), dtype=np.int64) if want_stats else None
    counters = (C.c_long * 4)()
    f = lib().orc_solve_ensemble_independent_fast
    f.restype = C.c_double
    secs = f(C.c_int(model), C.c_int(model_size), C.c_int(nsys), p.ctypes.data_as(_dp), C.c_int(np_), C.c_double(rtol), a_ptr, C.c_int(a_arr.size),
             C.c_double(t0), C.c_double(h0), C.c_double(t_final), C.c_int(nthreads), y.ctypes.data_as(_dp) if want_y else None,
             stats.ctypes.data_as(C.POINTER(C.c_long)) if want_stats else None, counters)
    if secs < 0:
        raise ValueError("oracle_fast: model does not qualify (identity mass, no roots, n in (3, 4))")
    return dict(seconds=secs, steps=int(counters[0]), newton_iterations=int(counters[1]), lu_setups=int(counters[2]), failed=int(counters[3]), y=y,
                stats=stats)


def solve_dense_independent(model, p, t_eval, *, model_size=0, rtol=1e-6, atol=(1e-6,), t0=0.0, h0=1.0, method=METHOD_BDF, nthreads=1, group=1, options=None):
    """solve_dense per member (each its own IVP; group > 1: consecutive groups of `group` members as one lock-step batched problem each); a member
    that finds a root stops there (its next column is the state at the root, later columns NaN; see solve_dense_independent.last_roots).  Returns y [nsys, nt, n], stats [nsys, 5] (steps, newton its, LU setups, error fails, newton fails), nfailed."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    nsys, np_ = p.shape
    a_arr, a_ptr = _d(np.asarray(atol, dtype=np.float64).reshape(-1))
    te, te_ptr = _d(t_eval)
    s = OracleSolver(model, p[0], model_size=model_size, rtol=rtol, atol=atol, t0=t0, h0=h0, method=method, options=options)
    n = s.n
    del s
    for k, v in (options or {}).items():  # problem.<ode / ic>_options.<k> = v for every problem of this call
        lib().orc_next_solver_option(k.encode(), C.c_double(v))
    y = np.empty((nsys, te.size, n))
    stats = np.zeros((nsys, 5), dtype=np.int64)
    root_t = np.full(nsys, np.nan)
    root_idx = np.full(nsys, -1, dtype=np.int32)
    ncols = np.zeros(nsys, dtype=np.int32)
    f = lib().orc_solve_dense_independent
    f.restype = C.c_int
    failed = f(C.c_int(model), C.c_int(model_size), C.c_int(nsys), p.ctypes.data_as(_dp), C.c_int(np_), C.c_double(rtol), a_ptr, C.c_int(a_arr.size),
               C.c_double(t0), C.c_double(h0), C.c_int(method), te_ptr, C.c_int(te.size), C.c_int(nthreads), C.c_int(group), y.ctypes.data_as(_dp),
               stats.ctypes.data_as(C.POINTER(C.c_long)), root_t.ctypes.data_as(_dp), root_idx.ctypes.data_as(C.POINTER(C.c_int)),
               ncols.ctypes.data_as(C.POINTER(C.c_int)))
    solve_dense_independent.last_roots = dict(t_root=root_t, root_idx=root_idx, ncols=ncols)
    return y, stats, int(failed)


def solve_dense_independent_sens(model, p, t_eval, *, model_size=0, rtol=1e-6, atol=(1e-6,), t0=0.0, h0=1.0, method=METHOD_BDF, sens_rtol=None, sens_atol=None,
                                 nthreads=1, group=1, options=None):
    """solve_dense_sensitivities per member (group > 1: per lock-step group of `group` members): BDF / TR-BDF2 / ESDIRK34 with forward sensitivities (problem.bdf_sens(), .tr_bdf2_sens(), .esdirk34_sens());
    sens_atol None: turn_off_sensitivities_error_control.  Returns y [nsys, nt, n], sens [np, nsys, nt, n], stats [nsys, 5], nfailed."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    nsys, np_ = p.shape
    a_arr, a_ptr = _d(np.asarray(atol, dtype=np.float64).reshape(-1))
    sa = np.zeros(0) if sens_atol is None else np.asarray(sens_atol, dtype=np.float64).reshape(-1)
    sa_buf, sa_ptr = _d(sa if sa.size else np.zeros(1))
    te, te_ptr = _d(t_eval)
    s = OracleSolver(model, p[0], model_size=model_size, rtol=rtol, atol=atol, t0=t0, h0=h0)
    n = s.n
    del s
    for k, v in (options or {}).items():
        lib().orc_next_solver_option(k.encode(), C.c_double(v))
    y = np.empty((nsys, te.size, n))
    sens = np.full((np_, nsys, te.size, n), np.nan)
    stats = np.zeros((nsys, 5), dtype=np.int64)
    f = lib().orc_solve_dense_independent_sens_method
    f.restype = C.c_int
    failed = f(C.c_int(model), C.c_int(model_size), C.c_int(nsys), p.ctypes.data_as(_dp), C.c_int(np_), C.c_double(rtol), a_ptr, C.c_int(a_arr.size), C.c_double(t0),
               C.c_double(h0), C.c_int(method), te_ptr, C.c_int(te.size), C.c_int(nthreads), C.c_int(group), C.c_double(0.0 if sens_rtol is None else sens_rtol), sa_ptr, C.c_int(sa.size),
               y.ctypes.data_as(_dp), sens.ctypes.data_as(_dp), stats.ctypes.data_as(C.POINTER(C.c_long)))
    return y, sens, stats, int(failed)


def set_det_pow(on):
    """Switch the oracle's pow() between libm (default: the reference's arithmetic) and include/diffsol_detpow.h (bit-comparable with the
    device-resident kernels run with deterministic_pow=True)."""
    lib().orc_set_det_pow(C.c_int(1 if on else 0))


def event_counts(reset=True):
    """Diagnostic counters of the BDF restatement since the last reset: how often a run takes each path (pow calls by site, step-size updates, order
    selections): what scripts/phase_frequencies.py weights the per-phase instruction counts of the device kernel with."""
    out = (C.c_long * 8)()
    lib().orc_event_counts(out, C.c_int(1 if reset else 0))
    names = ["pow_calls", "pow_first_iter", "pow_first_iter_eta_reset", "pow_first_iter_eta_reset_ts", "pow_rate", "step_size_updates", "order_selections", "powi_calls"]
    return dict(zip(names, list(out)))


def det_pow(x, y):
    f = lib().orc_det_pow
    f.restype = C.c_double
    return f(C.c_double(x), C.c_double(y))


def compute_r(order, factor):
    out = np.empty((order + 1) * (order + 1))
    lib().orc_compute_r(order, factor, out.ctypes.data_as(_dp))
    return out.reshape(order + 1, order + 1).T.copy()  # column-major -> [i, j]


def lu_solve(a, b):
    """a: [nbatch, n, n] (row, col), b: [nbatch, n]. Returns x, lu [nbatch, n, n], piv [nbatch, n], singular flag."""
    a = np.asarray(a, dtype=np.float64)
    nb, n, _ = a.shape
    a_cm = np.ascontiguousarray(np.transpose(a, (0, 2, 1)))
    x = np.ascontiguousarray(b, dtype=np.float64).copy()
    lu = np.empty_like(a_cm)
    piv = np.empty((nb, n), dtype=np.int32)
    rc = lib().orc_lu_solve(n, nb, a_cm.ctypes.data_as(_dp), x.ctypes.data_as(_dp), lu.ctypes.data_as(_dp), piv.ctypes.data_as(_ip))
    return x, np.transpose(lu, (0, 2, 1)).copy(), piv, rc


def lu_solve_fullpiv(a, b):
    """Complete-pivoting LU (the algorithm of the reference's FaerLU). a: [nbatch, n, n] (row, col), b: [nbatch, n]. Returns x, singular flag."""
    a = np.asarray(a, dtype=np.float64)
    nb, n, _ = a.shape
    a_cm = np.ascontiguousarray(np.transpose(a, (0, 2, 1)))
    x = np.ascontiguousarray(b, dtype=np.float64).copy()
    rc = lib().orc_lu_solve_fullpiv(n, nb, a_cm.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return x, rc


def squared_norm(x, y, atol, rtol):
    x = np.ascontiguousarray(x, dtype=np.float64)
    nb, n = x.shape
    xa, xp = _d(x)
    ya, yp = _d(y)
    aa, ap = _d(atol)
    return lib().orc_squared_norm(n, nb, xp, yp, ap, rtol)


def convergence_trace(norms, rtol=1e-6, tol=0.2, max_iter=10):
    na, npx = _d(norms)
    status = np.empty(na.size, dtype=np.int32)
    eta = np.empty(na.size)
    lib().orc_convergence_trace(rtol, tol, max_iter, npx, na.size, status.ctypes.data_as(_ip), eta.ctypes.data_as(_dp))
    return status, eta


def model_rhs(model, x, p, t=0.0, model_size=0):
    xa, xp = _d(x)
    pa, pp = _d(p)
    y = np.empty_like(xa)
    lib().orc_model_rhs(model, model_size, xp, pp, t, y.ctypes.data_as(_dp))
    return y


def model_jac_mul(model, x, p, v, t=0.0, model_size=0):
    xa, xp = _d(x)
    pa, pp = _d(p)
    va, vp = _d(v)
    y = np.empty_like(xa)
    lib().orc_model_jac_mul(model, model_size, xp, pp, t, vp, y.ctypes.data_as(_dp))
    return y


def model_sens_mul(model, x, p, v, t=0.0, model_size=0):
    """(dF/dp) v, v in parameter space; None for a model without parameter sensitivities."""
    x_a, x_p = _d(np.asarray(x, dtype=np.float64))
    p_a, p_p = _d(np.asarray(p, dtype=np.float64))
    v_a, v_p = _d(np.asarray(v, dtype=np.float64))
    y = np.empty(x_a.size)
    r = lib().orc_model_sens_mul(C.c_int(model), C.c_int(model_size), x_p, p_p, C.c_double(t), v_p, y.ctypes.data_as(_dp))
    return y if r == 0 else None


def model_init_sens_mul(model, p, v, n, t=0.0, model_size=0):
    """(du0/dp) v, v in parameter space (n = number of states); None for a model without parameter sensitivities."""
    p_a, p_p = _d(np.asarray(p, dtype=np.float64))
    v_a, v_p = _d(np.asarray(v, dtype=np.float64))
    y = np.empty(int(n))
    r = lib().orc_model_init_sens_mul(C.c_int(model), C.c_int(model_size), p_p, C.c_double(t), v_p, y.ctypes.data_as(_dp))
    return y if r == 0 else None


def model_root(model, x, p, t=0.0, model_size=0, max_roots=4):
    xa, xp = _d(x)
    pa, pp = _d(p)
    g = np.zeros(max_roots)
    k = lib().orc_model_root(model, model_size, xp, pp, t, g.ctypes.data_as(_dp))
    return g[:k]


DET_FN = {"exp": 0, "log": 1, "tanh": 2, "asinh": 3, "sin": 4, "cos": 5}


def det_fn(name, x):
    """include/diffsol_detpow.h's elementary functions (the ones the RLC / single-particle registry models are written with)."""
    f = lib().orc_det_fn
    return np.array([f(DET_FN[name], float(v)) for v in np.atleast_1d(x)])


def load_external_model(so_path):
    """Register a CPU model library with the external-model C ABI (dsl_dims, dsl_rhs, ...; generated from DiffSL by the product's front end and
    compiled by the test) and return its oracle model id."""
    mid = lib().orc_load_external_model(str(so_path).encode())
    if mid < 0:
        raise OracleError(f"cannot load external model {so_path}")
    return mid


def model_dims(model, model_size=0):
    out = (C.c_int * 5)()
    lib().orc_model_dims(model, model_size, out)
    return dict(n=out[0], nparams=out[1], nroots=out[2], nout=out[3], has_mass=bool(out[4]))


def model_init(model, p, t=0.0, model_size=0):
    pa, pp = _d(p)
    y = np.empty(model_dims(model, model_size)["n"])
    lib().orc_model_init(model, model_size, pp, t, y.ctypes.data_as(_dp))
    return y


def model_mass_gemv(model, x, p, y, beta, t=0.0, model_size=0):
    xa, xp = _d(x)
    pa, pp = _d(p)
    out = np.array(y, dtype=np.float64)
    lib().orc_model_mass_gemv(model, model_size, xp, pp, t, beta, out.ctypes.data_as(_dp))
    return out


def model_out(model, x, p, t=0.0, model_size=0):
    xa, xp = _d(x)
    pa, pp = _d(p)
    g = np.zeros(max(model_dims(model, model_size)["nout"], 1))
    k = lib().orc_model_out(model, model_size, xp, pp, t, g.ctypes.data_as(_dp))
    return g[:k]
