"""Python handle on the host-side integrators (include/diffsol_hip_solver.h): the harness tests and bench.py drive
`OdeBuilder ... .bdf() / .tr_bdf2() / .esdirk34()` through, mirroring the reference's user API
(crates/diffsol/src/ode_solver/builder.rs, problem.rs, method.rs)."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import DiffsolHipError, DshsOptions, check, vp

METHOD_BDF, METHOD_TR_BDF2, METHOD_ESDIRK34 = 0, 1, 2

MODELS = {
    "exponential_decay": 0, "exponential_decay_with_algebraic": 1, "exponential_decay_with_algebraic_batched": 2, "robertson_ode": 3,
    "robertson": 4, "dydt_y2": 5, "gaussian_decay": 6, "heat1d": 7, "rlc": 8, "exponential_decay_with_root": 9, "spm": 10,
    "heat2d": 11, "foodweb": 12,
}

STAT_NAMES = [
    "number_of_linear_solver_setups", "number_of_steps", "number_of_error_test_failures", "number_of_nonlinear_solver_iterations",
    "number_of_nonlinear_solver_fails", "number_of_linear_solver_setups_from_checkpoint",
    "number_of_linear_solver_setups_from_first_convergence_fail", "number_of_linear_solver_setups_from_second_convergence_fail",
    "number_of_linear_solver_setups_from_error_test_fail", "number_of_linear_solver_setups_from_step_success", "number_of_calls",
    "number_of_jac_muls", "number_of_matrix_evals",
]

STOP_INTERNAL_TIMESTEP, STOP_ROOT_FOUND, STOP_TSTOP_REACHED = 0, 1, 2
# dshs_set_ensemble_mode (include/diffsol_hip_solver.h): which integrator solve_dense runs for an ensemble
ENSEMBLE_AUTO, ENSEMBLE_LOCKSTEP, ENSEMBLE_PER_MEMBER, ENSEMBLE_WAVEFRONT = -1, 0, 1, 64


ARITH_EXACT, ARITH_FAST = 1, 2


def set_resident_arithmetic(mode):
    """dshs_set_resident_arithmetic: ARITH_FAST (library default) | ARITH_EXACT — the arithmetic of the device-resident BDF behind Solver.solve_dense in its
    default ensemble modes (include/diffsol_hip_solver.h).  The bitwise parity tier runs with ARITH_EXACT (tests/conftest.py: DSH_RESIDENT_ARITH=exact)."""
    check(_ffi.load_host_lib().dshs_set_resident_arithmetic(int(mode)), host=True)


def get_resident_arithmetic():
    return int(_ffi.load_host_lib().dshs_get_resident_arithmetic())


def set_deterministic_pow(on):
    """dshs_set_deterministic_pow: pow() of the host-driven integrators = include/diffsol_detpow.h (the device-resident integrators' pow) instead of libm."""
    check(_ffi.load_host_lib().dshs_set_deterministic_pow(1 if on else 0), host=True)


class Solver:
    """One ensemble solver instance on one GPU."""

    def __init__(self, model, p, *, nbatch=1, model_size=0, rtol=1e-6, atol=(1e-6,), t0=0.0, h0=1.0, method=METHOD_BDF, device=0, stream=None,
                 fused=True, block_threads=0, options=None, ensemble_mode=None, sens=False, sens_rtol=None, sens_atol=None):
        L = _ffi.load_host_lib()
        self._L = L
        if isinstance(model, str):
            model = MODELS[model]
        elif hasattr(model, "model_id"):  # diffsl.DiffslModel: a run-time-compiled model (kept alive by this solver)
            self._jit_model = model
            if p is None:
                p = np.tile(model.defaults, (nbatch, 1))
            model, model_size = model.model_id, 0
        p = np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape(-1))
        a = np.ascontiguousarray(np.asarray(atol, dtype=np.float64).reshape(-1))
        o = DshsOptions()
        L.dshs_default_options(C.byref(o))
        o.use_fused_kernels = 1 if fused else 0
        o.block_threads = int(block_threads)
        for k, v in (options or {}).items():
            setattr(o, k, v)
        h = vp()
        if sens:  # problem.bdf_sens(): forward sensitivities integrated alongside (sens_atol None: not part of the error control)
            sa = np.zeros(0) if sens_atol is None else np.ascontiguousarray(np.asarray(sens_atol, dtype=np.float64).reshape(-1))
            sa_buf = sa if sa.size else np.zeros(1)
            rc = L.dshs_create_sens(device, stream, model, model_size, nbatch, p.ctypes.data_as(_ffi.c_dp), p.size, rtol, a.ctypes.data_as(_ffi.c_dp), a.size,
                                    t0, h0, method, C.byref(o), 1, 0.0 if sens_rtol is None else float(sens_rtol), sa_buf.ctypes.data_as(_ffi.c_dp), sa.size, C.byref(h))
        else:
            rc = L.dshs_create(device, stream, model, model_size, nbatch, p.ctypes.data_as(_ffi.c_dp), p.size, rtol, a.ctypes.data_as(_ffi.c_dp), a.size,
                               t0, h0, method, C.byref(o), C.byref(h))
        check(rc, host=True)
        self._h = h
        self.n = int(L.dshs_nstates(h))
        self.nbatch = int(L.dshs_nbatch(h))
        self.fused = bool(L.dshs_is_fused(h))
        if ensemble_mode is not None:
            self.set_ensemble_mode(ensemble_mode)

    def set_ensemble_mode(self, mode):
        """ENSEMBLE_AUTO (default): solve_dense runs device-resident whenever the model has such a kernel; ENSEMBLE_LOCKSTEP: host-driven trait path."""
        check(self._L.dshs_set_ensemble_mode(self._h, int(mode)), host=True)

    def ensemble_mode(self):
        """(requested, resolved) ensemble mode of solve_dense."""
        a, b = C.c_int(), C.c_int()
        check(self._L.dshs_get_ensemble_mode(self._h, C.byref(a), C.byref(b)), host=True)
        return a.value, b.value

    def last_solve_info(self):
        """(mode the last solve_dense ran in, counters summed over the ensemble members)."""
        m, tot = C.c_int(), (C.c_int64 * 6)()
        check(self._L.dshs_last_solve_info(self._h, C.byref(m), tot), host=True)
        return m.value, dict(zip(self.ADAPTIVE_TOTALS, [int(v) for v in tot]))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.dshs_destroy(self._h)
            self._h = None

    def reset(self):
        """Re-create the solver state from the (device-resident) problem: a fresh `.bdf()` / `.tr_bdf2()` / `.esdirk34()`."""
        check(self._L.dshs_reset(self._h), host=True)

    def context_handle(self):
        """raw dsh_ctx* of this solver (dshs_context): what dist.CabiCommunicator takes"""
        return vp(self._L.dshs_context(self._h))

    def set_linear_solve_mode(self, mode):
        """0 = exact (default: the reference's order of operations, bit-identical to the CPU path); 1 = reordered (opt-in: the chunked-affine banded solve, ~1e-13 relative)"""
        check(self._L.dshs_set_linear_solve_mode(self._h, int(mode)), host=True)

    def set_kernel_timing(self, enable=True):
        check(self._L.dshs_set_kernel_timing(self._h, 1 if enable else 0), host=True)

    def set_kernel_timing_target(self, target):
        """which launches the event brackets go around: TIMING_RESIDENT (default), TIMING_LU_SOLVE, TIMING_LU_FACTOR (include/diffsol_hip.h DSH_TIMING_*)"""
        check(self._L.dshs_set_kernel_timing_target(self._h, int(target)), host=True)

    def kernel_timing(self):
        n, ms = C.c_int64(), C.c_double()
        check(self._L.dshs_get_kernel_timing(self._h, C.byref(n), C.byref(ms)), host=True)
        return int(n.value), float(ms.value)

    def kernel_timing_overhead_ms(self):
        ms, clk = C.c_double(), C.c_double()
        check(self._L.dshs_get_kernel_timing_overhead(self._h, C.byref(ms), C.byref(clk)), host=True)
        return float(ms.value), float(clk.value)

    def step(self):
        r = C.c_int()
        check(self._L.dshs_step(self._h, C.byref(r)), host=True)
        return r.value

    def set_stop_time(self, t):
        check(self._L.dshs_set_stop_time(self._h, t), host=True)

    def state(self):
        t, h, order = C.c_double(), C.c_double(), C.c_int()
        y = np.empty((self.nbatch, self.n))
        dy = np.empty((self.nbatch, self.n))
        check(self._L.dshs_get_state(self._h, C.byref(t), C.byref(h), C.byref(order), y.ctypes.data_as(_ffi.c_dp), dy.ctypes.data_as(_ffi.c_dp)), host=True)
        return dict(t=t.value, h=h.value, order=order.value, y=y, dy=dy)

    def scalars(self):
        t, h, order = C.c_double(), C.c_double(), C.c_int()
        check(self._L.dshs_get_state(self._h, C.byref(t), C.byref(h), C.byref(order), None, None), host=True)
        return t.value, h.value, order.value

    def diff(self):
        out = np.empty((self.nbatch, 8, self.n))
        check(self._L.dshs_bdf_get_diff(self._h, out.ctypes.data_as(_ffi.c_dp)), host=True)
        return out

    def interpolate(self, t):
        y = np.empty((self.nbatch, self.n))
        check(self._L.dshs_interpolate(self._h, t, y.ctypes.data_as(_ffi.c_dp)), host=True)
        return y

    def interpolate_sens(self, t=None):
        """OdeSolverMethod::interpolate_sens: [nparams, nbatch, n]; t=None: state.s at the current time."""
        npar = int(self._L.dshs_nparams(self._h))
        out = np.empty((npar, self.nbatch, self.n))
        check(self._L.dshs_interpolate_sens(self._h, float("nan") if t is None else float(t), out.ctypes.data_as(_ffi.c_dp)), host=True)
        return out

    def root_info(self):
        t, i = C.c_double(), C.c_int()
        self._L.dshs_root_info(self._h, C.byref(t), C.byref(i))
        return t.value, i.value

    def stats(self):
        out = (C.c_int64 * 13)()
        self._L.dshs_stats(self._h, out)
        return dict(zip(STAT_NAMES, [int(v) for v in out]))

    def solve_to_points(self, t_points):
        tp = np.ascontiguousarray(t_points, dtype=np.float64)
        out = np.empty((tp.size, self.nbatch, self.n))
        rc = check(self._L.dshs_solve_to_points(self._h, tp.ctypes.data_as(_ffi.c_dp), tp.size, out.ctypes.data_as(_ffi.c_dp)), host=True)
        return out, rc

    def solve(self, t_final, keep_trajectory=False):
        y = np.empty((self.nbatch, self.n))
        ncols, reason = C.c_int64(), C.c_int()
        check(self._L.dshs_solve(self._h, t_final, 1 if keep_trajectory else 0, y.ctypes.data_as(_ffi.c_dp), C.byref(ncols), C.byref(reason)), host=True)
        if keep_trajectory:
            ts = np.empty(ncols.value)
            ys = np.empty((ncols.value, self.nbatch, self.n))
            check(self._L.dshs_trajectory(self._h, ts.ctypes.data_as(_ffi.c_dp), ys.ctypes.data_as(_ffi.c_dp)), host=True)
            return y, int(ncols.value), reason.value, ts, ys
        return y, int(ncols.value), reason.value

    def solve_dense(self, t_eval, want_host=True, dev_ptr=None):
        te = np.ascontiguousarray(t_eval, dtype=np.float64)
        out = np.empty((te.size, self.nbatch, self.n)) if want_host else None
        reason = C.c_int()
        check(self._L.dshs_solve_dense(self._h, te.ctypes.data_as(_ffi.c_dp), te.size, out.ctypes.data_as(_ffi.c_dp) if want_host else None,
                                       vp(dev_ptr) if dev_ptr else None, C.byref(reason)), host=True)
        return out, reason.value


    ADAPTIVE_TOTALS = ["number_of_steps", "number_of_nonlinear_solver_iterations", "number_of_linear_solver_setups", "number_of_error_test_failures",
                       "number_of_nonlinear_solver_fails", "failed_members"]

    def solve_dense_adaptive(self, t_eval, want_host=True, dev_ptr=None, want_member_stats=False, group=1, deterministic_pow=True):
        """solve_dense with device-resident step-size/order control, the whole ensemble in ONE launch (dshs_solve_dense_adaptive):
        group=1 every member its own history and event time (CPU semantics of a sweep), group=64 wavefront-sized lock-step groups (batched
        semantics, nbatch 64).  deterministic_pow=True (default) uses the pow() of include/diffsol_detpow.h: bit-identical to the oracle in the same mode; False: ocml's pow().
        Returns (y [nt, nbatch, n] or None, totals dict[, member dict(stats [5, nbatch], status, t_root, root_idx, ncols)])."""
        te = np.ascontiguousarray(t_eval, dtype=np.float64)
        out = np.empty((te.size, self.nbatch, self.n)) if want_host else None
        totals = (C.c_int64 * 6)()
        m = None
        if want_member_stats:
            m = dict(stats=np.empty((5, self.nbatch), dtype=np.int32), status=np.empty(self.nbatch, dtype=np.int32), t_root=np.empty(self.nbatch),
                     root_idx=np.empty(self.nbatch, dtype=np.int32), ncols=np.empty(self.nbatch, dtype=np.int32))
        i32 = lambda a: a.ctypes.data_as(_ffi.c_i32p)
        check(self._L.dshs_solve_dense_adaptive(self._h, te.ctypes.data_as(_ffi.c_dp), te.size, int(group), int(deterministic_pow),
                                                out.ctypes.data_as(_ffi.c_dp) if want_host else None,
                                                vp(dev_ptr) if dev_ptr else None, i32(m["stats"]) if m else None, i32(m["status"]) if m else None,
                                                m["t_root"].ctypes.data_as(_ffi.c_dp) if m else None, i32(m["root_idx"]) if m else None,
                                                i32(m["ncols"]) if m else None, totals), host=True)
        tot = dict(zip(self.ADAPTIVE_TOTALS, [int(v) for v in totals]))
        return (out, tot, m) if want_member_stats else (out, tot)

    def solve_adaptive(self, t_final, max_cols=1024, group=1, deterministic_pow=True):
        """OdeSolverMethod::solve (method.rs:227-258) on the device-resident BDF (dshs_solve_adaptive): the state of every member after EVERY accepted step, the whole
        ensemble in one launch.  Returns (y [max_cols, nbatch, n], t [max_cols, nbatch], member dict(ncols, stats [5, nbatch], status, t_root, root_idx), totals dict);
        member b's solution is y[:ncols[b], b], t[:ncols[b], b] (column 0 = the initial state, the last one at t_final or at the member's event); ncols[b] > max_cols:
        call again with more room."""
        nb = self.nbatch
        y = np.full((max_cols, nb, self.n), np.nan)
        t = np.full((max_cols, nb), np.nan)
        m = dict(ncols=np.empty(nb, dtype=np.int32), stats=np.empty((5, nb), dtype=np.int32), status=np.empty(nb, dtype=np.int32), t_root=np.empty(nb),
                 root_idx=np.empty(nb, dtype=np.int32))
        totals = (C.c_int64 * 6)()
        i32 = lambda a: a.ctypes.data_as(_ffi.c_i32p)
        check(self._L.dshs_solve_adaptive(self._h, float(t_final), int(max_cols), int(group), int(deterministic_pow), y.ctypes.data_as(_ffi.c_dp), t.ctypes.data_as(_ffi.c_dp),
                                          i32(m["ncols"]), i32(m["stats"]), i32(m["status"]), m["t_root"].ctypes.data_as(_ffi.c_dp), i32(m["root_idx"]), totals), host=True)
        return y, t, m, dict(zip(self.ADAPTIVE_TOTALS, [int(v) for v in totals]))

    def solve_dense_adaptive_sens(self, t_eval, group=1, deterministic_pow=True, want_member_stats=False):
        """solve_dense_sensitivities (sensitivities.rs:114-260) on the device-resident BDF with forward sensitivities (dshs_solve_dense_adaptive_sens): states AND
        the sensitivities of every parameter at t_eval from one launch; the solver must have been created with sens=True.
        Returns (y [nt, nbatch, n], sens [nparams, nt, nbatch, n], totals dict[, member dict(stats [5, nbatch], status)])."""
        te = np.ascontiguousarray(t_eval, dtype=np.float64)
        npar = int(self._L.dshs_nparams(self._h))
        out = np.empty((te.size, self.nbatch, self.n))
        sens = np.empty((npar, te.size, self.nbatch, self.n))
        totals = (C.c_int64 * 6)()
        m = dict(stats=np.empty((5, self.nbatch), dtype=np.int32), status=np.empty(self.nbatch, dtype=np.int32)) if want_member_stats else None
        i32 = lambda a: a.ctypes.data_as(_ffi.c_i32p)
        check(self._L.dshs_solve_dense_adaptive_sens(self._h, te.ctypes.data_as(_ffi.c_dp), te.size, int(group), int(deterministic_pow),
                                                     out.ctypes.data_as(_ffi.c_dp), sens.ctypes.data_as(_ffi.c_dp), i32(m["stats"]) if m else None,
                                                     i32(m["status"]) if m else None, totals), host=True)
        tot = dict(zip(self.ADAPTIVE_TOTALS, [int(v) for v in totals]))
        return (out, sens, tot, m) if want_member_stats else (out, sens, tot)


class OdeBuilder:
    """OdeBuilder (crates/diffsol/src/ode_solver/builder.rs:22-146): fluent problem description; `.bdf()` etc. create the solver."""

    def __init__(self):
        self._kw = dict(rtol=1e-6, atol=(1e-6,), t0=0.0, h0=1.0, nbatch=1, model_size=0)
        self._model = None
        self._p = []

    def t0(self, v): self._kw["t0"] = v; return self
    def h0(self, v): self._kw["h0"] = v; return self
    def rtol(self, v): self._kw["rtol"] = v; return self
    def atol(self, v): self._kw["atol"] = tuple(np.atleast_1d(v)); return self
    def p(self, v): self._p = v; return self
    def nbatch(self, v): self._kw["nbatch"] = v; return self
    def device(self, v): self._kw["device"] = v; return self
    def stream(self, v): self._kw["stream"] = v; return self
    def fused(self, v): self._kw["fused"] = v; return self
    def options(self, **kw): self._kw["options"] = kw; return self

    def model(self, name, size=0):
        self._model = name
        self._kw["model_size"] = size
        return self

    def bdf(self): return Solver(self._model, self._p, method=METHOD_BDF, **self._kw)
    def tr_bdf2(self): return Solver(self._model, self._p, method=METHOD_TR_BDF2, **self._kw)
    def esdirk34(self): return Solver(self._model, self._p, method=METHOD_ESDIRK34, **self._kw)
