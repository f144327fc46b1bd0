"""The OPT-IN reordered banded solve (dsh_ctx_set_solve_mode(DSH_SOLVE_REORDERED), csrc/dsh_lu_band_affine.hpp): chunked affine maps instead of
the sequential chain — never the default, not bit-comparable, held to a tolerance against the exact solve (which is bit-identical to the oracle):
tridiagonal systems with and without row interchanges, sizes that are / are not multiples of the 16 chunks, ensembles that do not fill the last workgroup, a
singular system reported like the exact solve does, and BASELINE config 3 end to end (heat1d n = 512, TR-BDF2) to 1e-9 of the exact run with the same step counts."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


def _tridiag(rng, nb, n, pivoting):
    a = np.zeros((nb, n, n))
    for b in range(nb):
        d = rng.uniform(2.5, 4.0, n) if not pivoting else rng.uniform(0.2, 1.5, n)   # weak diagonal: partial pivoting interchanges rows
        lo, up = rng.uniform(-1.0, 1.0, n - 1), rng.uniform(-1.0, 1.0, n - 1)
        if pivoting:
            lo = lo * 2.0
        a[b] = np.diag(d) + np.diag(lo, -1) + np.diag(up, 1)
    return a


@pytest.mark.parametrize("n,nb,pivoting", [(512, 40, False), (512, 40, True), (100, 17, True), (33, 5, True), (1000, 6, True), (640, 3, False)])
def test_reordered_banded_solve_agrees_with_the_exact_solve(H, n, nb, pivoting):
    from diffsol_amd import _ffi
    rng = np.random.default_rng(n + nb)
    ctx = H.HipContext(0, nbatch=nb)
    L = _ffi.load_device_lib()
    a = _tridiag(rng, nb, n, pivoting)
    A = H.HipMat.from_array(a, ctx)
    rhs = rng.standard_normal((nb, n))
    lu = H.HipLU(ctx, n)
    lu.factor(A)
    assert lu.band_width() == 1
    x_exact = H.HipVec.from_vec(rhs, ctx)
    lu.solve_in_place(x_exact)
    xe = x_exact.clone_as_vec()
    assert L.dsh_ctx_get_solve_mode(ctx._h) == 0
    _ffi.check(L.dsh_ctx_set_solve_mode(ctx._h, 1))
    x_fast = H.HipVec.from_vec(rhs, ctx)
    lu.solve_in_place(x_fast)
    xf = x_fast.clone_as_vec()
    _ffi.check(L.dsh_ctx_set_solve_mode(ctx._h, 0))
    # the exact solve against numpy first (it is the bitwise-tested one), then the reordered against the exact
    ref = np.stack([np.linalg.solve(a[b], rhs[b]) for b in range(nb)])
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(xe - ref) / scale).max() < 1e-9
    assert (np.abs(xf - xe) / scale).max() < (1e-10 if pivoting else 1e-12)
    assert not np.array_equal(xf, xe) or n < 64  # it IS another order of operations
    if pivoting:  # the case was meant to interchange rows: the same operand through the dense LU (banded factors are not downloadable)
        lud = H.HipLU(ctx, n)
        lud.set_structure(True)
        lud.factor(A)
        assert (lud.factors()[1] != np.arange(n)[None, :]).any()


def test_reordered_solve_reports_a_zero_pivot_like_the_exact_solve(H):
    from diffsol_amd import _ffi
    ctx = H.HipContext(0, nbatch=4)
    L = _ffi.load_device_lib()
    n = 64
    a = _tridiag(np.random.default_rng(1), 4, n, False)
    a[2, 10:, :] = 0.0  # rows of zeros: a zero pivot in system 2
    a[2, :, 10:] = 0.0
    A = H.HipMat.from_array(a, ctx)
    lu = H.HipLU(ctx, n)
    lu.factor(A)
    for mode in (0, 1):
        _ffi.check(L.dsh_ctx_set_solve_mode(ctx._h, mode))
        x = H.HipVec.from_vec(np.ones((4, n)), ctx)
        with pytest.raises(H.DiffsolHipError) as e:
            lu.solve_in_place(x)
        assert "zero pivot in 1 system" in str(e.value)


def test_config_3_with_the_reordered_solve_stays_within_1e_9_of_the_exact_run(H):
    nb, n = 256, 512
    D = np.random.default_rng(12345).uniform(0.5, 2.0, (nb, 1))
    kw = dict(nbatch=nb, model_size=n, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
    s = H.Solver("heat1d", D, **kw)
    y_exact, _ = s.solve_to_points([0.5])
    st_exact = s.stats()
    s2 = H.Solver("heat1d", D, **kw)
    s2.set_linear_solve_mode(1)
    y_fast, _ = s2.solve_to_points([0.5])
    st_fast = s2.stats()
    assert st_fast["number_of_steps"] == st_exact["number_of_steps"] and st_fast["number_of_nonlinear_solver_iterations"] == st_exact["number_of_nonlinear_solver_iterations"]
    assert np.abs(y_fast - y_exact).max() < 1e-9
