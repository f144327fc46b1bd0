"""ctypes binding of include/diffsol_c_hip.h — what pydiffsol is to crates/diffsol-c: `Ode(code)` ... `ode.solve(params, t_final)`.
Test / example harness only; every call goes through the C ABI of libdiffsol_hip_host.so."""
import ctypes as C

import numpy as np

from . import _ffi

OK, ERR, BAD_ARG = 0, -1, -2
MATRIX_HIP_DENSE = 3
LINEAR_SOLVER_DEFAULT, LINEAR_SOLVER_LU, LINEAR_SOLVER_KLU = 0, 1, 2
ODE_SOLVER_BDF, ODE_SOLVER_ESDIRK34, ODE_SOLVER_TR_BDF2, ODE_SOLVER_TSIT45 = 0, 1, 2, 3
SCALAR_F64 = 1
JIT_HIPRTC = 2
ENSEMBLE_AUTO, ENSEMBLE_LOCKSTEP, ENSEMBLE_PER_MEMBER, ENSEMBLE_WAVEFRONT = -1, 0, 1, 64

_vp, _i32, _sz, _dbl, _dp = C.c_void_p, C.c_int32, C.c_size_t, C.c_double, C.POINTER(C.c_double)
_ODE_OPTS = [("max_nonlinear_solver_iterations", _sz), ("max_error_test_failures", _sz), ("update_jacobian_after_steps", _sz), ("update_rhs_jacobian_after_steps", _sz),
             ("threshold_to_update_jacobian", _dbl), ("threshold_to_update_rhs_jacobian", _dbl), ("min_timestep", _dbl)]
_IC_OPTS = [("use_linesearch", _i32), ("max_linesearch_iterations", _sz), ("max_newton_iterations", _sz), ("max_linear_solver_setups", _sz),
            ("step_reduction_factor", _dbl), ("armijo_constant", _dbl)]

C_ABI = {
    "diffsol_error_code": (_i32, []), "diffsol_error": (C.c_char_p, []), "diffsol_last_error_message": (C.c_char_p, []),
    "diffsol_last_error_file": (C.c_char_p, []), "diffsol_last_error_line": (C.c_uint32, []), "diffsol_clear_last_error": (None, []),
    "diffsol_host_array_alloc_vector": (_vp, [_sz, _i32]), "diffsol_host_array_free": (None, [_vp]), "diffsol_host_array_ptr": (_vp, [_vp]),
    "diffsol_host_array_ndim": (_sz, [_vp]), "diffsol_host_array_dim": (_sz, [_vp, _sz]), "diffsol_host_array_stride": (_sz, [_vp, _sz]),
    "diffsol_host_array_dtype": (_i32, [_vp]),
    "diffsol_ode_new_jit": (_vp, [C.c_char_p, _i32, _i32, _i32, _i32]), "diffsol_ode_free": (None, [_vp]),
    "diffsol_ode_get_options": (_i32, [_vp, C.POINTER(_vp)]), "diffsol_ode_get_ic_options": (_i32, [_vp, C.POINTER(_vp)]),
    "diffsol_ode_y0": (_i32, [_vp, _dp, _sz, C.POINTER(_vp)]), "diffsol_ode_rhs": (_i32, [_vp, _dp, _sz, _dbl, _dp, _sz, C.POINTER(_vp)]),
    "diffsol_ode_rhs_jac_mul": (_i32, [_vp, _dp, _sz, _dbl, _dp, _sz, _dp, _sz, C.POINTER(_vp)]),
    "diffsol_ode_solve": (_i32, [_vp, _dp, _sz, _dbl, C.POINTER(_vp)]), "diffsol_ode_solve_dense": (_i32, [_vp, _dp, _sz, _dp, _sz, C.POINTER(_vp)]),
    "diffsol_ode_solve_fwd_sens": (_i32, [_vp, _dp, _sz, _dp, _sz, C.POINTER(_vp)]),
    "diffsol_ode_get_sens_rtol": (_i32, [_vp, C.POINTER(_i32), _dp]), "diffsol_ode_set_sens_rtol": (_i32, [_vp, _i32, _dbl]),
    "diffsol_ode_get_sens_atol": (_i32, [_vp, C.POINTER(_i32), _dp]), "diffsol_ode_set_sens_atol": (_i32, [_vp, _i32, _dbl]),
    "diffsol_solution_wrapper_get_sens": (_i32, [_vp, C.POINTER(C.POINTER(_vp)), C.POINTER(_sz)]), "diffsol_host_array_list_free": (None, [C.POINTER(_vp), _sz]),
    "diffsol_ode_new_external": (_vp, [_i32, _i32, _i32, _vp, _sz, _vp, _sz, _vp, _sz]),
    "diffsol_ode_new_external_dynamic": (_vp, [C.c_char_p, _i32, _i32, _i32, _vp, _sz, _vp, _sz, _vp, _sz]),
    "diffsol_ode_get_integrate_out": (_i32, [_vp, C.POINTER(_i32)]), "diffsol_ode_set_integrate_out": (_i32, [_vp, _i32]),
    "diffsol_alloc_string": (_vp, [_sz]), "diffsol_free_string": (None, [_vp, _sz]), "diffsol_alloc": (_vp, [_sz, _sz]), "diffsol_free": (None, [_vp, _sz, _sz]),
    "diffsol_ode_get_matrix_type": (_i32, [_vp]), "diffsol_ode_get_ode_solver": (_i32, [_vp]), "diffsol_ode_set_ode_solver": (_i32, [_vp, _i32]),
    "diffsol_ode_get_linear_solver": (_i32, [_vp]), "diffsol_ode_set_linear_solver": (_i32, [_vp, _i32]),
    "diffsol_ode_get_ensemble_mode": (_i32, [_vp]), "diffsol_ode_set_ensemble_mode": (_i32, [_vp, _i32]),
    "diffsol_ode_get_dims": (_i32, [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "diffsol_ode_set_atol_vector": (_i32, [_vp, _dp, _sz]),
    "diffsol_ode_options_free": (None, [_vp]), "diffsol_ic_options_free": (None, [_vp]),
    "diffsol_solution_wrapper_free": (None, [_vp]), "diffsol_solution_wrapper_get_ys": (_i32, [_vp, C.POINTER(_vp)]),
    "diffsol_solution_wrapper_get_ts": (_i32, [_vp, C.POINTER(_vp)]),
    "diffsol_solution_wrapper_get_member_info": (C.c_int64, [_vp, C.POINTER(_i32), _dp, C.POINTER(_i32), C.POINTER(_i32)]),
}
for _kind in ("matrix", "linear_solver", "ode_solver", "scalar", "jit_backend"):
    C_ABI[f"diffsol_{_kind}_type_count"] = (_sz, [])
    C_ABI[f"diffsol_{_kind}_type_is_valid"] = (_i32, [_i32])
    C_ABI[f"diffsol_{_kind}_type_name"] = (C.c_char_p, [_i32])
for _f in ("out_rtol", "out_atol", "param_rtol", "param_atol"):
    C_ABI[f"diffsol_ode_get_{_f}"] = (_i32, [_vp, C.POINTER(_i32), _dp])
    C_ABI[f"diffsol_ode_set_{_f}"] = (_i32, [_vp, _i32, _dbl])
for _f in ("rtol", "atol", "t0", "h0"):
    C_ABI[f"diffsol_ode_get_{_f}"] = (_i32, [_vp, _dp])
    C_ABI[f"diffsol_ode_set_{_f}"] = (_i32, [_vp, _dbl])
for _prefix, _fields in (("diffsol_ode_options", _ODE_OPTS), ("diffsol_ic_options", _IC_OPTS)):
    for _name, _ty in _fields:
        C_ABI[f"{_prefix}_get_{_name}"] = (_i32, [_vp, C.POINTER(_ty)])
        C_ABI[f"{_prefix}_set_{_name}"] = (_i32, [_vp, _ty])

_lib = None


def lib():
    global _lib
    if _lib is None:
        _ffi.load_device_lib()
        _lib = _ffi._bind(C.CDLL(_ffi.lib_paths()[1]), C_ABI)
    return _lib


class DiffsolCError(RuntimeError):
    pass


def _check(rc):
    if rc != OK:
        L = lib()
        msg = L.diffsol_last_error_message()
        raise DiffsolCError(f"[{rc}] {(msg or b'').decode(errors='replace')} ({(L.diffsol_last_error_file() or b'').decode()}:{L.diffsol_last_error_line()})")


def _to_numpy(arr):
    """HostArray -> numpy copy honouring shape and byte strides; frees the handle."""
    L = lib()
    try:
        nd = L.diffsol_host_array_ndim(arr)
        shape = [L.diffsol_host_array_dim(arr, i) for i in range(nd)]
        strides = [L.diffsol_host_array_stride(arr, i) for i in range(nd)]
        n = int(np.prod(shape)) if shape else 0
        if n == 0:
            return np.empty(shape)
        buf = (C.c_double * n).from_address(L.diffsol_host_array_ptr(arr))
        flat = np.frombuffer(buf, dtype=np.float64)
        return np.lib.stride_tricks.as_strided(flat, shape=shape, strides=strides).copy()
    finally:
        L.diffsol_host_array_free(arr)


class _Options:
    def __init__(self, handle, prefix, fields, free):
        object.__setattr__(self, "_h", handle)
        object.__setattr__(self, "_prefix", prefix)
        object.__setattr__(self, "_fields", dict(fields))
        object.__setattr__(self, "_free", free)

    def __getattr__(self, name):
        ty = self._fields[name]
        out = ty()
        _check(getattr(lib(), f"{self._prefix}_get_{name}")(self._h, C.byref(out)))
        return out.value

    def __setattr__(self, name, value):
        _check(getattr(lib(), f"{self._prefix}_set_{name}")(self._h, self._fields[name](value)))

    def __del__(self):
        if self._h:
            self._free(self._h)


class Solution:
    def __init__(self, handle):
        self._h = handle

    def _array(self, fn):
        out = _vp()
        _check(fn(self._h, C.byref(out)))
        return _to_numpy(out)

    @property
    def ys(self):
        return self._array(lib().diffsol_solution_wrapper_get_ys)

    @property
    def ts(self):
        return self._array(lib().diffsol_solution_wrapper_get_ts)

    @property
    def sens(self):
        """diffsol_solution_wrapper_get_sens: one array per parameter, shaped like ys (empty list unless the solution comes from solve_fwd_sens)."""
        L = lib()
        lst, n = C.POINTER(_vp)(), _sz()
        _check(L.diffsol_solution_wrapper_get_sens(self._h, C.byref(lst), C.byref(n)))
        out = [_to_numpy(_vp(lst[j])) for j in range(n.value)]
        if n.value:
            L.diffsol_host_array_list_free(lst, n.value)
        return out

    def member_info(self):
        L = lib()
        nb = L.diffsol_solution_wrapper_get_member_info(self._h, None, None, None, None)
        status, idx, cols, t_root = np.zeros(nb, np.int32), np.zeros(nb, np.int32), np.zeros(nb, np.int32), np.zeros(nb)
        L.diffsol_solution_wrapper_get_member_info(self._h, status.ctypes.data_as(C.POINTER(_i32)), t_root.ctypes.data_as(_dp), idx.ctypes.data_as(C.POINTER(_i32)),
                                                   cols.ctypes.data_as(C.POINTER(_i32)))
        return dict(status=status, t_root=t_root, root_index=idx, ncols=cols)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().diffsol_solution_wrapper_free(self._h)
            self._h = None


class Ode:
    """pydiffsol-style handle: Ode(code, ode_solver=...) -> solve / solve_dense / y0 / rhs / rhs_jac_mul, properties rtol, atol, t0, h0, ..."""

    def __init__(self, code, matrix_type=MATRIX_HIP_DENSE, linear_solver=LINEAR_SOLVER_DEFAULT, ode_solver=ODE_SOLVER_BDF, jit_backend=JIT_HIPRTC):
        h = lib().diffsol_ode_new_jit(code.encode(), jit_backend, matrix_type, linear_solver, ode_solver)
        if not h:
            _check(ERR)
        self._h = h

    @classmethod
    def external_dynamic(cls, path, matrix_type=MATRIX_HIP_DENSE, linear_solver=LINEAR_SOLVER_DEFAULT, ode_solver=ODE_SOLVER_BDF):
        """diffsol_ode_new_external_dynamic: a HIP source file with the reference's external model functions as device functions (include/diffsol_c_hip.h)."""
        h = lib().diffsol_ode_new_external_dynamic(str(path).encode(), matrix_type, linear_solver, ode_solver, None, 0, None, 0, None, 0)
        if not h:
            _check(ERR)
        self = cls.__new__(cls)
        self._h = h
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            lib().diffsol_ode_free(self._h)
            self._h = None

    @staticmethod
    def _p(a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
        return a, a.ctypes.data_as(_dp), a.size

    def dims(self):
        a, b, c, d = _sz(), _sz(), _sz(), _sz()
        _check(lib().diffsol_ode_get_dims(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(nstates=a.value, nparams=b.value, nout=c.value, nroots=d.value)

    def y0(self, params):
        p, pp, n = self._p(params)
        out = _vp()
        _check(lib().diffsol_ode_y0(self._h, pp, n, C.byref(out)))
        return _to_numpy(out)

    def rhs(self, params, t, y):
        p, pp, n = self._p(params)
        ya, yp, yn = self._p(y)
        out = _vp()
        _check(lib().diffsol_ode_rhs(self._h, pp, n, t, yp, yn, C.byref(out)))
        return _to_numpy(out)

    def rhs_jac_mul(self, params, t, y, v):
        p, pp, n = self._p(params)
        ya, yp, yn = self._p(y)
        va, vpp, vn = self._p(v)
        out = _vp()
        _check(lib().diffsol_ode_rhs_jac_mul(self._h, pp, n, t, yp, yn, vpp, vn, C.byref(out)))
        return _to_numpy(out)

    def solve(self, params, final_time):
        p, pp, n = self._p(params)
        out = _vp()
        _check(lib().diffsol_ode_solve(self._h, pp, n, final_time, C.byref(out)))
        return Solution(out)

    def solve_dense(self, params, t_eval):
        p, pp, n = self._p(params)
        te, tp, tn = self._p(t_eval)
        out = _vp()
        _check(lib().diffsol_ode_solve_dense(self._h, pp, n, tp, tn, C.byref(out)))
        return Solution(out)

    def solve_fwd_sens(self, params, t_eval):
        """diffsol_ode_solve_fwd_sens: states and forward sensitivities at t_eval (Solution.ys, Solution.sens)."""
        p, pp, n = self._p(params)
        te, tp, tn = self._p(t_eval)
        out = _vp()
        _check(lib().diffsol_ode_solve_fwd_sens(self._h, pp, n, tp, tn, C.byref(out)))
        return Solution(out)

    def _opt_get(self, fn):
        some, val = _i32(), _dbl()
        _check(fn(self._h, C.byref(some), C.byref(val)))
        return val.value if some.value else None

    sens_rtol = property(lambda self: self._opt_get(lib().diffsol_ode_get_sens_rtol),
                         lambda self, v: _check(lib().diffsol_ode_set_sens_rtol(self._h, 0 if v is None else 1, 0.0 if v is None else float(v))))
    sens_atol = property(lambda self: self._opt_get(lib().diffsol_ode_get_sens_atol),
                         lambda self, v: _check(lib().diffsol_ode_set_sens_atol(self._h, 0 if v is None else 1, 0.0 if v is None else float(v))))

    def set_atol_vector(self, atol):
        a, ap, n = self._p(atol)
        _check(lib().diffsol_ode_set_atol_vector(self._h, ap, n))

    @property
    def options(self):
        out = _vp()
        _check(lib().diffsol_ode_get_options(self._h, C.byref(out)))
        return _Options(out, "diffsol_ode_options", _ODE_OPTS, lib().diffsol_ode_options_free)

    @property
    def ic_options(self):
        out = _vp()
        _check(lib().diffsol_ode_get_ic_options(self._h, C.byref(out)))
        return _Options(out, "diffsol_ic_options", _IC_OPTS, lib().diffsol_ic_options_free)


def _scalar_prop(name):
    def get(self):
        out = _dbl()
        _check(getattr(lib(), f"diffsol_ode_get_{name}")(self._h, C.byref(out)))
        return out.value

    def set_(self, v):
        _check(getattr(lib(), f"diffsol_ode_set_{name}")(self._h, float(v)))
    return property(get, set_)


def _enum_prop(name):
    def get(self):
        return getattr(lib(), f"diffsol_ode_get_{name}")(self._h)

    def set_(self, v):
        _check(getattr(lib(), f"diffsol_ode_set_{name}")(self._h, int(v)))
    return property(get, set_)


for _f in ("rtol", "atol", "t0", "h0"):
    setattr(Ode, _f, _scalar_prop(_f))
for _f in ("ode_solver", "linear_solver", "ensemble_mode"):
    setattr(Ode, _f, _enum_prop(_f))
Ode.matrix_type = property(lambda self: lib().diffsol_ode_get_matrix_type(self._h))
