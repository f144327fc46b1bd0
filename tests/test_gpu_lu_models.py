"""GPU parity tests: batched LU (row a5), model registry (row a16) and the fused kernels (rows a7, a9-a11) against the CPU oracle and
against compositions of the 1:1 trait ops.  Integer/bit-level equality wherever the arithmetic order is identical."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import ORACLE_MODEL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


@pytest.fixture(scope="module")
def ctx1(H):
    return H.HipContext(0, nbatch=1)


@pytest.mark.parametrize("n,nb", [(1, 5), (2, 2), (3, 1), (3, 1000), (4, 67), (5, 64), (8, 129), (9, 3), (12, 33), (16, 257), (17, 40), (31, 65), (32, 9), (33, 21), (42, 70), (48, 10), (49, 9), (64, 19), (65, 7), (100, 9), (137, 5), (138, 4), (150, 6), (256, 3), (300, 2), (512, 2), (600, 2), (1100, 1)])
def test_lu_factor_and_solve_match_oracle_bitwise(H, O, ctx1, n, nb):
    rng = np.random.default_rng(n * 100 + nb)
    c = ctx1.clone_with_nbatch(nb)
    a = rng.standard_normal((nb, n, n))
    a[:, 0, 0] *= 1e-6  # force pivoting
    b = rng.standard_normal((nb, n))
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    x = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x)
    x_ref, lu_ref, piv_ref, rc = O.lu_solve(a, b)
    assert rc == 0 and lu.n_singular() == 0
    got_lu, got_piv = lu.factors()
    assert np.array_equal(got_piv, piv_ref)
    assert np.array_equal(got_lu, lu_ref)
    assert np.array_equal(x.clone_as_vec(), x_ref)
    assert np.allclose(np.einsum("bij,bj->bi", a, x_ref), b, atol=1e-6)


@pytest.mark.parametrize("n,nb", [(65, 7), (100, 9), (300, 3), (512, 4), (700, 2), (962, 3), (1024, 2)])
def test_streaming_dense_solve_with_prefetched_factor_panels_gives_the_bits_of_the_blocked_solve(H, ctx1, monkeypatch, n, nb):
    """k_lu_solve_stream (a register ring of factor panels in flight, the interchanges applied from LDS visiting only the rows that move) performs per element the
    operations of k_lu_solve_blocked in their order: same bits for every ring depth; also with a matrix that needs no interchange at all."""
    rng = np.random.default_rng(n + nb)
    c = ctx1.clone_with_nbatch(nb)
    for dominant in (False, True):
        a = rng.standard_normal((nb, n, n))
        if dominant:
            a += 2.0 * n * np.eye(n)[None]  # no row moves
        else:
            a[:, 0, 0] *= 1e-6
        b = rng.standard_normal((nb, n))
        lu = H.HipLU(c, n)
        lu.factor(H.HipMat.from_array(a, c))
        sols = []
        for depth in ("0", "2", "4", "6"):
            monkeypatch.setenv("DSH_LU_STREAM_SOLVE", depth)
            x = H.HipVec.from_vec(b, c)
            lu.solve_in_place(x)
            sols.append(x.clone_as_vec())
        monkeypatch.delenv("DSH_LU_STREAM_SOLVE")
        assert all(np.array_equal(sols[0], q) for q in sols[1:])
        assert np.allclose(np.einsum("bij,bj->bi", a, np.asarray(sols[0]).reshape(nb, n)), b, atol=1e-6 * n)


def test_lu_reference_diagonal_kat_and_singular_reporting(H, ctx1):
    """2x2 diagonal solve incl. the batched variant (diffsol/src/linear_solver/mod.rs:283-321); zero pivot -> LuSolveFailed."""
    c2 = ctx1.clone_with_nbatch(2)
    a = np.array([[[2.0, 0.0], [0.0, 2.0]], [[4.0, 0.0], [0.0, 4.0]]])
    lu = H.HipLU(c2, 2)
    lu.factor(H.HipMat.from_array(a, c2))
    x = H.HipVec.from_vec([[2.0, 4.0], [2.0, 4.0]], c2)
    lu.solve_in_place(x)
    assert x.clone_as_vec().tolist() == [[1.0, 2.0], [0.5, 1.0]]
    a[1] = [[1.0, 2.0], [2.0, 4.0]]
    lu.factor(H.HipMat.from_array(a, c2))
    assert lu.n_singular() == 1
    with pytest.raises(H.DiffsolHipError) as e:
        lu.solve_in_place(H.HipVec.from_vec([[1.0, 1.0], [1.0, 1.0]], c2))
    assert e.value.code == -3
    unfactored = H.HipLU(c2, 2)
    with pytest.raises(H.DiffsolHipError) as e:
        unfactored.solve_in_place(H.HipVec.zeros(2, c2))
    assert e.value.code == -4  # LuNotInitialized


@pytest.mark.parametrize("n", [12, 150])
def test_lu_cooperative_kernels_report_singular_members(H, ctx1, n):
    """n > 8 goes through the wavefront-per-system (n=12) / workgroup-per-system blocked (n=150) kernels: same zero-pivot reporting as the register path."""
    nb = 5
    c = ctx1.clone_with_nbatch(nb)
    rng = np.random.default_rng(n)
    a = rng.standard_normal((nb, n, n))
    a[3, :, 4] = 0.0  # a zero column stays exactly zero through the elimination: exact zero pivot at step 4
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    assert lu.n_singular() == 1
    with pytest.raises(H.DiffsolHipError) as e:
        lu.solve_in_place(H.HipVec.from_vec(np.ones((nb, n)), c))
    assert e.value.code == -3
    a[3] = np.eye(n)
    lu.factor(H.HipMat.from_array(a, c))
    assert lu.n_singular() == 0
    x = H.HipVec.from_vec(np.ones((nb, n)), c)
    lu.solve_in_place(x)
    assert np.array_equal(x.clone_as_vec()[3], np.ones(n))


MODEL_CASES = [("exponential_decay", 0, 2), ("exponential_decay_with_algebraic", 0, 1), ("robertson_ode", 1, 3), ("robertson_ode", 3, 3), ("robertson", 0, 3),
               ("dydt_y2", 10, 0), ("gaussian_decay", 10, 10), ("heat1d", 16, 1), ("rlc", 0, 6), ("spm", 20, 1), ("spm", 5, 1),
               ("heat2d", 6, 1), ("heat2d", 10, 1), ("foodweb", 5, 2), ("foodweb", 10, 2)]


@pytest.mark.parametrize("name,size,np_", MODEL_CASES)
def test_model_rhs_jac_mul_jacobian_match_oracle(H, O, ctx1, name, size, np_):
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb = 37
    c = ctx1.clone_with_nbatch(nb)
    mid = H.MODELS[name]
    n64, p64 = C.c_int64(), C.c_int64()
    assert L.dsh_model_info(mid, size, C.byref(n64), C.byref(p64), None, None) == 0
    n = n64.value
    rng = np.random.default_rng(n + size)
    x, v = rng.uniform(0.1, 1.0, (nb, n)), rng.standard_normal((nb, n))
    p = rng.uniform(0.5, 2.0, (nb, max(np_, 1)))[:, :np_]
    t = 0.7
    X, Vv, Y = H.HipVec.from_vec(x, c), H.HipVec.from_vec(v, c), H.HipVec.zeros(n, c)
    P = H.HipVec.from_vec(p, c) if np_ else H.HipVec.zeros(0, c)
    assert L.dsh_model_rhs(c._h, mid, size, nb, t, X.ptr, P.ptr, Y.ptr) == 0
    ref = np.stack([O.model_rhs(ORACLE_MODEL[name], x[b], p[b], t, size) for b in range(nb)])
    assert np.array_equal(Y.clone_as_vec(), ref)  # rlc included: its sin() is include/diffsol_detpow.h's on both sides
    assert L.dsh_model_jac_mul(c._h, mid, size, nb, t, X.ptr, P.ptr, Vv.ptr, Y.ptr) == 0
    ref = np.stack([O.model_jac_mul(ORACLE_MODEL[name], x[b], p[b], v[b], t, size) for b in range(nb)])
    assert np.array_equal(Y.clone_as_vec(), ref)
    J = H.HipMat.zeros(n, n, c)
    assert L.dsh_model_jacobian(c._h, mid, size, nb, t, X.ptr, P.ptr, J.ptr) == 0
    jref = np.empty((nb, n, n))
    for b in range(nb):
        for j in range(n):
            e = np.zeros(n); e[j] = 1.0
            jref[b, :, j] = O.model_jac_mul(ORACLE_MODEL[name], x[b], p[b], e, t, size)
    assert np.array_equal(J.to_array(), jref)
    Y0 = H.HipVec.zeros(n, c)
    assert L.dsh_model_init(c._h, mid, size, nb, 0.0, P.ptr, Y0.ptr) == 0
    assert Y0.clone_as_vec().shape == (nb, n)
    nroots = len(O.model_root(ORACLE_MODEL[name], x[0], p[0], t, size))
    if nroots:  # stop conditions (spm: the terminal voltage, tanh / asinh / exp / sqrt): same bits as the oracle's
        if name == "spm":  # concentrations inside the physical range
            x = np.concatenate([rng.uniform(0.0, 1.0, (nb, 1)), rng.uniform(0.0, 1.0, (nb, 1)), rng.uniform(2e4, 4.5e4, (nb, size)), rng.uniform(2e3, 2e4, (nb, size))], axis=1)
            X = H.HipVec.from_vec(x, c)
        G = H.HipVec.zeros(nroots, c)
        assert L.dsh_model_root(c._h, mid, size, nb, t, X.ptr, P.ptr, G.ptr) == 0
        gref = np.stack([O.model_root(ORACLE_MODEL[name], x[b], p[b], t, size) for b in range(nb)])
        assert np.isfinite(gref).all() and np.array_equal(G.clone_as_vec(), gref)


def test_bdf_callable_kat_through_trait_ops_and_fused_kernel(H, ctx1, kats):
    """op/bdf.rs:318-361: F(y) and J = M - c f'(y) for exponential decay, c=0.1, psi_neg_y0=(1.1,1.2), y=(1,1): composed from the 1:1 ops
    exactly like BdfCallable::call_inplace / jacobian_inplace, and via the fused kernels."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    k = kats["bdf_callable_kat"]
    mid, c = H.MODELS[k["model"]], k["c"]
    y, psi, p = H.HipVec.from_vec(k["y"], ctx1), H.HipVec.from_vec(k["psi_neg_y0"], ctx1), H.HipVec.from_vec(k["p"], ctx1)
    out, tmp = H.HipVec.zeros(2, ctx1), H.HipVec.zeros(2, ctx1)
    L.dsh_model_rhs(ctx1._h, mid, 0, 1, k["t"], y.ptr, p.ptr, out.ptr)
    tmp.copy_from(y); tmp.add_assign(psi); out.axpy(1.0, tmp, -c)
    assert np.allclose(out.clone_as_vec()[0], k["F"], atol=k["tol"])
    J, A = H.HipMat.zeros(2, 2, ctx1), H.HipMat.zeros(2, 2, ctx1)
    L.dsh_model_jacobian(ctx1._h, mid, 0, 1, k["t"], y.ptr, p.ptr, J.ptr)
    A.scale_add_and_assign(H.HipMat.from_diagonal(H.HipVec.from_element(2, 1.0, ctx1)), -c, J)
    assert A.to_array()[0].tolist() == k["J"]
    # fused: jac_factor then one Newton iteration solves J delta = F, y -= delta
    lu = H.HipLU(ctx1, 2)
    rhs_jac = H.HipMat.zeros(2, 2, ctx1)
    assert L.dsh_jac_factor(ctx1._h, mid, 0, 1, k["t"], c, y.ptr, p.ptr, 1, rhs_jac.ptr, None, lu._h) == 0
    assert np.array_equal(rhs_jac.to_array(), J.to_array())
    atol = H.HipVec.from_vec([1e-6, 1e-6], ctx1)
    res = (C.c_double * 3)()
    ynew = y.clone()
    assert L.dsh_bdf_newton_iter(ctx1._h, mid, 0, 1, k["t"], c, y.ptr, ynew.ptr, psi.ptr, p.ptr, lu._h, y.ptr, y.ptr, atol.ptr, 1, 1e-6, res) == 0
    delta = np.array(k["F"]) / 1.01
    assert np.allclose(ynew.clone_as_vec()[0], np.array(k["y"]) - delta, atol=1e-12)
    w = np.abs(np.array(k["y"])) * 1e-6 + 1e-6
    assert np.isclose(res[0], np.mean((delta / w) ** 2), rtol=1e-12) and res[2] == 0.0


def test_sdirk_callable_and_robertson_jacobian_kats(H, ctx1, kats):
    """op/sdirk.rs:338-389 (F(k) = k - h f(phi + c k)) and :316-336 (dense Jacobian * v == jac_mul)."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    k = kats["sdirk_callable_kat"]
    mid = H.MODELS[k["model"]]
    y, phi, p = H.HipVec.from_vec(k["y"], ctx1), H.HipVec.from_vec(k["phi"], ctx1), H.HipVec.from_vec(k["p"], ctx1)
    tmp, out = phi.clone(), H.HipVec.zeros(2, ctx1)
    tmp.axpy(k["c"], y, 1.0)
    L.dsh_model_rhs(ctx1._h, mid, 0, 1, k["t"], tmp.ptr, p.ptr, out.ptr)
    out.axpy(1.0, y, -k["h"])
    assert np.allclose(out.clone_as_vec()[0], k["F"], atol=k["tol"])
    r = kats["sdirk_robertson_jacobian_kat"]
    mid = H.MODELS[r["model"]]
    y, phi, p, v = (H.HipVec.from_vec(r[q], ctx1) for q in ("y", "phi", "p", "v"))
    tmp = phi.clone()
    tmp.axpy(r["c"], y, 1.0)
    J, Mm, A = H.HipMat.zeros(3, 3, ctx1), H.HipMat.zeros(3, 3, ctx1), H.HipMat.zeros(3, 3, ctx1)
    L.dsh_model_jacobian(ctx1._h, mid, 0, 1, r["t"], tmp.ptr, p.ptr, J.ptr)
    L.dsh_model_mass_matrix(ctx1._h, mid, 0, 1, r["t"], p.ptr, Mm.ptr)
    assert Mm.to_array()[0].tolist() == [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]]
    A.scale_add_and_assign(Mm, -(r["c"] * r["h"]), J)
    Av = H.HipVec.zeros(3, ctx1)
    A.gemv(1.0, v, 0.0, Av)
    jv = H.HipVec.zeros(3, ctx1)
    L.dsh_model_jac_mul(ctx1._h, mid, 0, 1, r["t"], tmp.ptr, p.ptr, v.ptr, jv.ptr)
    L.dsh_model_mass_gemv(ctx1._h, mid, 0, 1, r["t"], v.ptr, p.ptr, -(r["c"] * r["h"]), jv.ptr)
    assert np.allclose(Av.clone_as_vec(), jv.clone_as_vec(), atol=r["tol"])


@pytest.mark.parametrize("order", [1, 2, 3, 5])
@pytest.mark.parametrize("n,nb", [(3, 1), (3, 130), (7, 65)])
def test_fused_bdf_prepare_and_accept_match_trait_op_composition(H, ctx1, order, n, nb):
    """dsh_bdf_prepare_step / dsh_bdf_accept_step vs the reference's op-by-op sequences (bdf.rs:568-577, :646-692, :1472-1478, :871-900)."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    rng = np.random.default_rng(order * 10 + n + nb)
    c = ctx1.clone_with_nbatch(nb)
    d = rng.standard_normal((nb, n, 8))
    gamma = np.concatenate([[0.0], np.cumsum(1.0 / np.arange(1, 6))])
    alpha = 0.77
    ru = rng.standard_normal((order + 1, order + 1))
    D, Dt = H.HipMat.from_array(d, c), H.HipMat.zeros(n, 8, c)
    yp, psi = H.HipVec.zeros(n, c), H.HipVec.zeros(n, c)
    ru_cm = np.ascontiguousarray(ru.T)
    assert L.dsh_bdf_prepare_step(c._h, n, nb, order, D.ptr, Dt.ptr, ru_cm.ctypes.data_as(_ffi.c_dp), gamma.ctypes.data_as(_ffi.c_dp), alpha, yp.ptr, psi.ptr) == 0
    nd = np.zeros_like(d)
    for j in range(order + 1):
        acc = d[:, :, 0] * ru[0, j]
        for kk in range(1, order + 1):
            acc = d[:, :, kk] * ru[kk, j] + acc
        nd[:, :, j] = acc
    assert np.array_equal(Dt.to_array()[:, :, : order + 1], nd[:, :, : order + 1])
    ypr = np.zeros((nb, n))
    for j in range(order + 1):
        ypr = ypr + nd[:, :, j]
    ps = gamma[1] * nd[:, :, 1]
    for j in range(2, order + 1):
        ps = gamma[j] * nd[:, :, j] + 1.0 * ps
    ps = ps * alpha - ypr
    assert np.array_equal(yp.clone_as_vec(), ypr) and np.array_equal(psi.clone_as_vec(), ps)
    # accept
    ynew = ypr + 1e-3 * rng.standard_normal((nb, n))
    atol = np.abs(rng.standard_normal(n)) * 1e-3 + 1e-6
    h, rtol = 0.37, 1e-4
    Ynew, Y, DY, AT = H.HipVec.from_vec(ynew, c), H.HipVec.zeros(n, c), H.HipVec.zeros(n, c), H.HipVec.from_vec(atol, ctx1)
    Dn = H.HipMat.from_array(nd, c)
    res = (C.c_double * 2)()
    psi_next = H.HipVec.zeros(n, c)
    assert L.dsh_bdf_accept_step(c._h, n, nb, order, h, Dn.ptr, yp.ptr, Ynew.ptr, Y.ptr, DY.ptr, AT.ptr, 1, rtol, gamma.ctypes.data_as(_ffi.c_dp), alpha,
                                 psi_next.ptr, 1, res) == 0
    e = nd.copy()
    dd = ynew - ypr
    e[:, :, order + 2] = dd - e[:, :, order + 1]
    e[:, :, order + 1] = dd
    for i in range(order, -1, -1):
        e[:, :, i] = e[:, :, i] + 1.0 * e[:, :, i + 1]
    assert np.array_equal(Dn.to_array(), e)
    assert np.array_equal(Y.clone_as_vec(), ypr) and np.array_equal(DY.clone_as_vec(), e[:, :, 1] * (1.0 / h))
    w = np.abs(ypr) * rtol + atol
    def sq(v):
        acc = np.zeros(nb)
        for i in range(n):
            term = v[:, i] / w[:, i]
            acc = acc + term * term
        return (acc / n).max()
    assert res[0] == sq(e[:, :, order]) and res[1] == sq(e[:, :, order + 2])
    # speculative prediction for the next step == what dsh_bdf_prepare_step computes from the updated D
    yp2, psi2 = H.HipVec.zeros(n, c), H.HipVec.zeros(n, c)
    assert L.dsh_bdf_prepare_step(c._h, n, nb, order, Dn.ptr, Dt.ptr, None, gamma.ctypes.data_as(_ffi.c_dp), alpha, yp2.ptr, psi2.ptr) == 0
    assert np.array_equal(yp.clone_as_vec(), yp2.clone_as_vec()) and np.array_equal(psi_next.clone_as_vec(), psi2.clone_as_vec())


def _banded(rng, nb, n, kl, ku, dominant):
    a = np.zeros((nb, n, n))
    for d in range(-kl, ku + 1):
        idx = np.arange(max(0, -d), min(n, n - d))
        a[:, idx, idx + d] = rng.standard_normal((nb, idx.size))
    if dominant:
        a[:, np.arange(n), np.arange(n)] += 4.0 * (kl + ku + 1)
    return a


@pytest.mark.parametrize("n,kl,ku", [(16, 1, 1), (42, 1, 1), (100, 2, 1), (64, 0, 3), (77, 3, 3), (512, 1, 1), (130, 4, 2), (33, 4, 4)])
@pytest.mark.parametrize("dominant", [True, False])
def test_banded_operands_in_dense_containers_are_solved_by_the_banded_kernels_with_the_bits_of_the_dense_lu(H, O, n, kl, ku, dominant):
    """dsh_lu_factor probes the bandwidth and eliminates only the band (LAPACK dgbtrf-style pivoting): the solution must equal, BIT FOR BIT, the dense
    kernels' and the oracle's dense partial-pivoting LU — with and without diagonal dominance (i.e. with real row interchanges and fill-in)."""
    nb = 70 if n < 500 else 9
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(1000 * n + 10 * kl + ku)
    a = _banded(rng, nb, n, kl, ku, dominant)
    b = rng.standard_normal((nb, n))
    A = H.HipMat.from_array(a, c)
    lu = H.HipLU(c, n)
    lu.factor(A)
    assert lu.band_width() == max(kl, ku) and lu.n_singular() == 0
    x = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x)
    xo, _, _, rc = O.lu_solve(a, b)
    assert rc == 0 and np.array_equal(x.clone_as_vec(), xo)
    dense = H.HipLU(c, n)
    dense.set_structure(True)
    dense.factor(A)
    assert dense.band_width() == 0
    xd = H.HipVec.from_vec(b, c)
    dense.solve_in_place(xd)
    assert np.array_equal(xd.clone_as_vec(), xo)
    with pytest.raises(H.DiffsolHipError):
        lu.factors()  # banded factors have no dense image
    # a second right-hand side, and re-factoring a wider operand with the same handle falls back to the dense kernels
    b2 = rng.standard_normal((nb, n))
    x2 = H.HipVec.from_vec(b2, c)
    lu.solve_in_place(x2)
    assert np.array_equal(x2.clone_as_vec(), O.lu_solve(a, b2)[0])
    wide = _banded(rng, nb, n, 5, 0, True)
    lu.factor(H.HipMat.from_array(wide, c))
    assert lu.band_width() == (5 if 2 * (2 * 5 + 0 + 1) <= n else 0)  # round 6: wider bands go to the general banded kernels where their factors pay (tests/test_gpu_lu_gband.py)
    x3 = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x3)
    assert np.array_equal(x3.clone_as_vec(), O.lu_solve(wide, b)[0])


_SPECIAL_BAND_SOLVE = """
import sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import diffsol_amd as H
from test_gpu_lu_models import _special_band_case
n, k, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
a, b = _special_band_case(n, k)
c = H.HipContext(nbatch=a.shape[0])
lu = H.HipLU(c, n)
lu.factor(H.HipMat.from_array(a, c))
assert lu.band_width() == k
x = H.HipVec.from_vec(b, c)
lu.solve_in_place(x)
np.save(out, np.asarray(x.clone_as_vec()).reshape(a.shape[0], n))
"""


def _special_band_case(n, k):
    nb = 37
    rng = np.random.default_rng(77 * n + k)
    a = _banded(rng, nb, n, k, k, True)
    a[3] *= 1e-150   # diagonal below 2^-300: every chunk of these systems takes the ordinary division
    a[11] *= 1e180
    a[20, 100:140] *= 1e-120  # a stretch of one system only
    b = rng.standard_normal((nb, n))
    b[0] = 0.0
    b[1, ::2] = -0.0
    b[2, 5:60] = 0.0
    b[4, ::7] = 1e-310
    b[5] *= 1e-200
    b[6] *= 1e250
    b[7, 300 % n] = -0.0
    b[8] = 0.0; b[8, n // 2] = 1.0   # a unit vector: zeros above it stay +0 through the backward sweep
    b[9] = -0.0; b[9, n - 1] = -1.0
    return a, b


@pytest.mark.parametrize("n,k", [(512, 1), (200, 2), (131, 4)])
def test_wide_banded_solve_splits_its_divisions_only_where_that_keeps_the_bits(O, tmp_path, n, k):
    """k_lu_band_solve_wide runs the division by U's diagonal as `refined reciprocal of the diagonal (off the chain) x numerator` (dsh_device.hpp:
    div_refined_rcp / div_by_refined — the instruction sequence of the compiler's own x / y when no operand needs scaling) and re-runs a chunk with
    ordinary divisions when an operand is outside that range.  Right-hand sides with +0, -0, denormal, tiny and huge entries, and systems scaled far out
    of range (mixed with ordinary ones in the same wavefront): the bit patterns — signs of zeros included — must be those of the one-lane-per-system
    kernel, which divides the ordinary way (DSH_LU_BAND_WIDE=0; the choice is read once per process, hence two child processes), and the values those of
    the oracle's dense LU (whose eliminations with structurally zero multipliers may turn a -0 into +0, so signs of zeros are not compared there)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _SPECIAL_BAND_SOLVE.format(root=root, tests=os.path.join(root, "tests"))
    res = {}
    for wide in ("0", "1"):
        out = str(tmp_path / f"x{wide}.npy")
        subprocess.run([sys.executable, "-c", code, str(n), str(k), out], check=True, env=dict(os.environ, DSH_LU_BAND_WIDE=wide), timeout=300)
        res[wide] = np.load(out)
    assert np.array_equal(res["0"].view(np.int64), res["1"].view(np.int64))
    a, b = _special_band_case(n, k)
    xo, _, _, rc = O.lu_solve(a, b)
    assert rc == 0 and np.array_equal(res["1"], np.asarray(xo).reshape(res["1"].shape))
    assert (res["1"] == 0.0).any() and (np.abs(res["1"]) > 1e100).any() and (np.abs(res["1"][res["1"] != 0.0]) < 1e-100).any()  # the case does reach the edges


def test_banded_lu_reports_singular_systems_like_the_dense_one(H, O):
    nb, n = 40, 30
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(3)
    a = _banded(rng, nb, n, 1, 2, True)
    a[7, :, 11] = 0.0  # a zero column: exact zero pivot at step 11 of system 7
    a[21, :, 0] = 0.0
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    assert lu.band_width() == 2 and lu.n_singular() == 2
    x = H.HipVec.from_vec(np.ones((nb, n)), c)
    with pytest.raises(H.DiffsolHipError) as e:
        lu.solve_in_place(x)
    assert "LuSolveFailed" in str(e.value) and "2 system" in str(e.value)
    ok = [b for b in range(nb) if b not in (7, 21)]
    xo = O.lu_solve(a[ok], np.ones((len(ok), n)))[0]
    assert np.array_equal(x.clone_as_vec()[ok], xo)


def test_declared_band_assembly_and_factorisation_touch_only_the_band_and_give_the_same_bits(H, O, ctx1):
    """dsh_mat_scale_add_assign_banded + dsh_lu_factor_banded: what the host-side integrators use when a model declares the structure of its Jacobian
    (dsh_model_band).  Entries outside the declared band are neither read nor written."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb, n, kl, ku = 33, 40, 2, 1
    c = ctx1.clone_with_nbatch(nb)
    rng = np.random.default_rng(8)
    jac, mass = _banded(rng, nb, n, kl, ku, False), _banded(rng, nb, n, 0, 0, True)
    J, M = H.HipMat.from_array(jac, c), H.HipMat.from_array(mass, c)
    poison = np.full((nb, n, n), 7.0)
    A = H.HipMat.from_array(poison, c)
    assert L.dsh_mat_scale_add_assign_banded(c._h, n, nb, kl, ku, A.ptr, M.ptr, nb, -0.3, J.ptr, nb) == 0
    got = A.to_array()
    i, j = np.indices((n, n))
    band = (i - j <= kl) & (j - i <= ku)
    assert np.array_equal(got[:, band], (jac * -0.3 + mass)[:, band]) and np.all(got[:, ~band] == 7.0)
    # factorisation with the declared band: nothing outside |i - j| <= max(kl, ku) is read (poison there), inside it the operand must be exact
    exact = jac * -0.3 + mass
    sym = np.abs(i - j) <= max(kl, ku)
    A2 = H.HipMat.from_array(np.where(sym, exact, 7.0), c)
    lu = H.HipLU(c, n)
    assert L.dsh_lu_factor_banded(lu._h, A2.ptr, kl, ku) == 0 and lu.band_width() == 2
    b = rng.standard_normal((nb, n))
    x = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x)
    assert np.array_equal(x.clone_as_vec(), O.lu_solve(jac * -0.3 + mass, b)[0])
    for model, size, expect in [("heat1d", 64, (1, 1, 0, 0)), ("spm", 20, (1, 1, 0, 0)), ("robertson_ode", 4, (2, 2, 0, 0)), ("gaussian_decay", 10, (0, 0, 0, 0)), ("rlc", 0, (-1, -1, -1, -1))]:
        out = [C.c_int() for _ in range(4)]
        assert L.dsh_model_band(H.MODELS[model], size, *[C.byref(o) for o in out]) == 0 and tuple(o.value for o in out) == expect


def _pack_band(a, kl, ku):
    """[nb, n, n] dense -> [nb, (kl + ku + 1) * n] band container: entry (i, j) at plane j - i + kl, row i; the corners outside the matrix are zeros"""
    nb, n, _ = a.shape
    out = np.zeros((nb, kl + ku + 1, n))
    for d in range(-kl, ku + 1):
        i = np.arange(max(0, -d), min(n, n - d))
        out[:, d + kl, i] = a[:, i, i + d]
    return out.reshape(nb, -1)


@pytest.mark.parametrize("n,kl,ku", [(16, 1, 1), (42, 1, 1), (100, 2, 1), (64, 0, 3), (77, 3, 3), (512, 1, 1), (130, 4, 2)])
@pytest.mark.parametrize("dominant", [True, False])
def test_band_containers_factor_solve_and_multiply_with_the_bits_of_the_dense_route(H, O, ctx1, n, kl, ku, dominant):
    """the band container ((kl + ku + 1) n entries per member instead of n^2; diffsol_hip.h dsh_mat_band_*, dsh_lu_create_banded,
    dsh_lu_factor_packed).  Factorisation + solve on a banded LU handle (factor storage (3k + 1) n per member) give the oracle's dense LU solution bit for
    bit, with and without real row interchanges; x + beta*y of containers is the entry-wise scale_add_and_assign; gemv and from_diagonal agree with the
    dense kernels; a dense operand is refused by the banded handle."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb = 70 if n < 500 else 9
    c = ctx1.clone_with_nbatch(nb)
    rng = np.random.default_rng(7000 * n + 10 * kl + ku)
    jac, mass = _banded(rng, nb, n, kl, ku, False), _banded(rng, nb, n, 0, 0, True)
    if dominant:
        mass[:, np.arange(n), np.arange(n)] += 4.0 * (kl + ku + 1)
    w = (kl + ku + 1) * n
    Jb, Mb = H.HipVec.from_vec(_pack_band(jac, kl, ku), c), H.HipVec.from_vec(_pack_band(mass, kl, ku), c)
    Ab = H.HipVec.from_vec(np.full((nb, w), 7.0), c)
    assert L.dsh_mat_scale_add_assign(c._h, w, nb, Ab.ptr, Mb.ptr, nb, -0.3, Jb.ptr, nb) == 0
    exact = jac * -0.3 + mass
    assert np.array_equal(Ab.clone_as_vec(), _pack_band(exact, kl, ku))
    h = C.c_void_p()
    k = max(1, kl, ku)
    assert L.dsh_lu_create_banded(c._h, n, nb, k, C.byref(h)) == 0
    try:
        assert L.dsh_lu_factor_packed(h, Ab.ptr, kl, ku) == 0 and L.dsh_lu_band_width(h) == k
        b = rng.standard_normal((nb, n))
        x = H.HipVec.from_vec(b, c)
        assert L.dsh_lu_solve(h, x.ptr) == 0
        xo, _, _, rc = O.lu_solve(exact, b)
        assert rc == 0 and np.array_equal(x.clone_as_vec(), xo)
        assert L.dsh_lu_factor(h, H.HipMat.from_array(exact, c).ptr) != 0  # no room for dense factors in this handle
    finally:
        L.dsh_lu_destroy(h)
    # the same container through an ordinary handle
    lu = H.HipLU(c, n)
    assert L.dsh_lu_factor_packed(lu._h, Ab.ptr, kl, ku) == 0
    x2 = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x2)
    assert np.array_equal(x2.clone_as_vec(), xo)
    # gemv: y = alpha A x + beta y, and from_diagonal
    xv, yv = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
    X, Y, Yd = H.HipVec.from_vec(xv, c), H.HipVec.from_vec(yv, c), H.HipVec.from_vec(yv, c)
    assert L.dsh_mat_band_gemv(c._h, n, nb, kl, ku, 1.7, Ab.ptr, X.ptr, nb, -0.4, Y.ptr) == 0
    assert L.dsh_mat_gemv(c._h, n, n, nb, 1.7, H.HipMat.from_array(exact, c).ptr, nb, X.ptr, nb, -0.4, Yd.ptr) == 0
    assert np.array_equal(Y.clone_as_vec(), Yd.clone_as_vec())
    D = H.HipVec.from_vec(np.full((nb, w), 3.0), c)
    assert L.dsh_mat_band_from_diagonal(c._h, n, nb, kl, ku, X.ptr, nb, D.ptr) == 0
    dd = np.zeros((nb, n, n)); dd[:, np.arange(n), np.arange(n)] = xv
    assert np.array_equal(D.clone_as_vec(), _pack_band(dd, kl, ku))


def test_band_container_jacobian_of_a_declared_model_holds_the_entries_of_the_dense_one(H, ctx1):
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb, n = 9, 64
    c = ctx1.clone_with_nbatch(nb)
    rng = np.random.default_rng(5)
    x, p = H.HipVec.from_vec(rng.uniform(0.1, 1.0, (nb, n)), c), H.HipVec.from_vec(rng.uniform(0.5, 2.0, (nb, 1)), c)
    dense = H.HipMat.from_array(np.zeros((nb, n, n)), c)
    assert L.dsh_model_jacobian(c._h, H.MODELS["heat1d"], n, nb, 0.0, x.ptr, p.ptr, dense.ptr) == 0
    for kl, ku in ((1, 1), (2, 3)):
        band = H.HipVec.from_vec(np.full((nb, (kl + ku + 1) * n), 7.0), c)
        assert L.dsh_model_jacobian_band_packed(c._h, H.MODELS["heat1d"], n, nb, 0.0, x.ptr, p.ptr, kl, ku, band.ptr) == 0
        assert np.array_equal(band.clone_as_vec(), _pack_band(dense.to_array(), kl, ku))
    assert L.dsh_model_jacobian_band_packed(c._h, H.MODELS["heat1d"], n, nb, 0.0, x.ptr, p.ptr, 0, 1, band.ptr) != 0  # narrower than the declared band


@pytest.mark.parametrize("n,nrhs", [(3, 5), (8, 3), (42, 4), (100, 2)])
def test_lu_solve_with_several_right_hand_sides_equals_separate_solves_bitwise(H, O, ctx1, n, nrhs):
    """dsh_lu_solve_multi: the factors serve nrhs columns per system (forward-sensitivity solves); same bits as one dsh_lu_solve per column and as the oracle."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb = 70
    c = ctx1.clone_with_nbatch(nb)
    rng = np.random.default_rng(n * nrhs)
    a = rng.standard_normal((nb, n, n)) + 3.0 * np.eye(n)
    rhs = rng.standard_normal((nb, n, nrhs))
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    B = H.HipMat.from_array(rhs, c)
    assert L.dsh_lu_solve_multi(lu._h, B.ptr, nrhs) == 0
    got = B.to_array()
    for r in range(nrhs):
        assert np.array_equal(got[:, :, r], O.lu_solve(a, rhs[:, :, r])[0])


@pytest.mark.parametrize("n,nb", [(144, 5), (256, 3), (512, 4)])
def test_opt_in_matrix_core_trailing_update_agrees_with_the_bit_exact_factorisation_to_tolerance(H, ctx1, monkeypatch, n, nb):
    """DSH_LU_MFMA=1: the trailing update of the blocked dense LU (n > 137, n a multiple of 16) on v_mfma_f64_16x16x4_f64 (dsh_lu_coop.hpp).  The matrix
    cores fuse the multiply-adds, so this path is NOT bit-identical — north_star's bar for floating point is 1e-6 relative.  Held here far tighter on
    well-conditioned systems: same pivot rows, factors to 1e-11 of the largest entry, solutions to 1e-9 relative, residual at rounding level; and the
    default path (no environment variable) is untouched — bitwise equal to what it was before the call."""
    rng = np.random.default_rng(n)
    c = ctx1.clone_with_nbatch(nb)
    a = rng.standard_normal((nb, n, n)) + np.eye(n) * 3.0
    a[:, 0, 0] *= 1e-6  # force at least one interchange
    b = rng.standard_normal((nb, n))
    out = {}
    for flag in ("0", "1", "0"):
        monkeypatch.setenv("DSH_LU_MFMA", flag)
        lu = H.HipLU(c, n)
        lu.factor(H.HipMat.from_array(a, c))
        x = H.HipVec.from_vec(b, c)
        lu.solve_in_place(x)
        f, p = lu.factors()
        out.setdefault(flag, []).append((f, p, x.clone_as_vec()))
    (f0, p0, x0), (f0b, p0b, x0b) = out["0"]
    f1, p1, x1 = out["1"][0]
    assert np.array_equal(f0, f0b) and np.array_equal(p0, p0b) and np.array_equal(x0, x0b)
    assert not np.array_equal(f0, f1), "the matrix-core path did not run (factors are bitwise those of the vector path)"
    assert np.array_equal(p0, p1)
    assert np.max(np.abs(f1 - f0)) <= 1e-11 * np.max(np.abs(f0))
    assert np.max(np.abs(x1 - x0)) <= 1e-9 * np.max(np.abs(x0))
    assert np.max(np.abs(np.einsum("bij,bj->bi", a, x1) - b)) <= 1e-10 * n


@pytest.mark.parametrize("n,nb", [(256, 6), (257, 3), (272, 5), (288, 5), (289, 3), (300, 4), (352, 7), (448, 3), (496, 9), (512, 6), (513, 2), (600, 3), (962, 2), (1024, 2)])
@pytest.mark.parametrize("kind", ["random", "dominant"])
def test_default_matrix_core_lu_keeps_the_pivots_and_agrees_with_the_oracle_to_rounding(H, O, ctx1, monkeypatch, n, nb, kind):
    """The default dense LU for 288 <= n <= 1024 (dsh_lu_tiled.hpp: row-major working copy, rows never move, register-resident panels, U12 and the trailing
    update on v_mfma_f64_16x16x4_f64) replaces CudaLU's host loop over cusolverDnDgetrf (linear_solver/cuda/lu.rs:59-125).  Fused multiply-adds and the
    matrix cores' summation order make it differ from the exact kernels in the last bits; north_star's bar for floating point is 1e-6 relative.  Held here
    to: the SAME pivot sequence as the oracle's partial-pivoting LU (random matrices: an interchange at almost every step), factors within 1e-11 of the
    largest entry, solutions within 1e-9 relative, residual at rounding level — and the exact mode (DSH_LU_EXACT=1) still gives the oracle's bits."""
    rng = np.random.default_rng(7 * n + nb)
    c = ctx1.clone_with_nbatch(nb)
    a = rng.standard_normal((nb, n, n))
    if kind == "dominant":
        a += np.eye(n) * (2.0 * np.sqrt(n))
        a[:, 0, 0] *= 1e-7  # one forced interchange
    b = rng.standard_normal((nb, n))
    x_ref, lu_ref, piv_ref, rc = O.lu_solve(a, b)
    assert rc == 0
    monkeypatch.delenv("DSH_LU_EXACT", raising=False)
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    x = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x)
    assert lu.n_singular() == 0
    f, p = lu.factors()
    xs = x.clone_as_vec()
    assert np.array_equal(p, piv_ref)
    assert not np.array_equal(f, lu_ref), "the matrix-core kernel did not run (factors are bitwise the exact kernel's)"
    assert np.max(np.abs(f - lu_ref)) <= 1e-11 * np.max(np.abs(lu_ref))
    assert np.max(np.abs(xs - x_ref)) <= 1e-9 * np.max(np.abs(x_ref)) * (1 if kind == "dominant" else n)
    assert np.max(np.abs(np.einsum("bij,bj->bi", a, xs) - b)) <= 1e-10 * n * max(1.0, np.max(np.abs(xs)))
    monkeypatch.setenv("DSH_LU_EXACT", "1")
    lu.factor(H.HipMat.from_array(a, c))
    f2, p2 = lu.factors()
    assert np.array_equal(f2, lu_ref) and np.array_equal(p2, piv_ref)


def test_default_matrix_core_lu_reports_singular_systems_and_leaves_zero_columns_alone(H, O, ctx1, monkeypatch):
    """A zero pivot column in one system of the ensemble: counted by dsh_lu_info, pivots[k] = k and no elimination at that step (what the exact kernels and
    the oracle do), the other systems unaffected; solving then fails with LuSolveFailed like the reference's getrs loop."""
    monkeypatch.delenv("DSH_LU_EXACT", raising=False)
    n, nb = 300, 4
    rng = np.random.default_rng(5)
    c = ctx1.clone_with_nbatch(nb)
    a = rng.standard_normal((nb, n, n))
    a[2, :, 100] = 0.0  # a zero column: the reduced matrix has one at step 100 too
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    assert lu.n_singular() == 1
    f, p = lu.factors()
    _, lu_ref, piv_ref, _ = O.lu_solve(a, rng.standard_normal((nb, n)))
    assert np.array_equal(p, piv_ref) and p[2, 100] == 100
    assert np.max(np.abs(f - lu_ref)) <= 1e-11 * np.max(np.abs(lu_ref))
    with pytest.raises(H.DiffsolHipError) as e:
        lu.solve_in_place(H.HipVec.from_vec(rng.standard_normal((nb, n)), c))
    assert e.value.code == -3


@pytest.mark.parametrize("name,size,n,npar", [("heat1d", 64, 64, 1), ("spm", 20, 42, 1), ("robertson_ode", 8, 24, 3)])
def test_band_only_jacobian_evaluation_writes_the_bits_of_the_dense_one(H, ctx1, name, size, n, npar):
    """dsh_model_jacobian_band: the Jacobian of a run-time-sized registry model on its declared band only, into a zeroed container = dsh_model_jacobian's dense
    result bit for bit (what the host-driven integrators now do for such models: config 3 writes 50 MB per evaluation instead of 8.6 GB); models without a
    declared band or with a static form are refused."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb = 37
    c = ctx1.clone_with_nbatch(nb)
    rng = np.random.default_rng(n)
    x, p = rng.uniform(0.1, 0.9, (nb, n)), rng.uniform(0.5, 1.5, (nb, npar))
    X, P = H.HipVec.from_vec(x, c), H.HipVec.from_vec(p, c)
    import ctypes as C
    kl, ku, ml, mu = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert L.dsh_model_band(H.MODELS[name], size, C.byref(kl), C.byref(ku), C.byref(ml), C.byref(mu)) == 0 and kl.value >= 0
    assert L.dsh_model_has_band_jacobian(H.MODELS[name], size) == 1
    Jd, Jb = H.HipMat.zeros(n, n, c), H.HipMat.zeros(n, n, c)
    assert L.dsh_model_jacobian(c._h, H.MODELS[name], size, nb, 0.3, X.ptr, P.ptr, Jd.ptr) == 0
    assert L.dsh_model_jacobian_band(c._h, H.MODELS[name], size, nb, 0.3, X.ptr, P.ptr, kl.value, ku.value, Jb.ptr) == 0
    d, b = Jd.to_array(), Jb.to_array()
    assert np.array_equal(d, b) and np.abs(d).max() > 0
    i, j = np.indices((n, n))
    assert not d[:, (j - i > ku.value) | (i - j > kl.value)].any()
    assert L.dsh_model_has_band_jacobian(H.MODELS["robertson_ode"], 1) == 0 and L.dsh_model_has_band_jacobian(H.MODELS["rlc"], 0) == 0  # register-resident forms
    assert L.dsh_model_jacobian_band(c._h, H.MODELS["robertson_ode"], 1, nb, 0.0, X.ptr, P.ptr, 2, 2, Jb.ptr) < 0
    assert L.dsh_model_jacobian_band(c._h, H.MODELS[name], size, nb, 0.0, X.ptr, P.ptr, kl.value - 1, ku.value, Jb.ptr) < 0  # narrower than the declared band


@pytest.mark.parametrize("n,k,nb", [(130, 1, 5), (512, 1, 16), (512, 1, 37), (300, 2, 9), (640, 1, 3), (900, 1, 4), (256, 3, 6), (100, 10, 9), (300, 21, 5)])
@pytest.mark.parametrize("ynb_full", [True, False])
def test_banded_solve_with_the_norm_fused_into_its_launch_gives_the_bits_of_solve_then_norm(H, O, n, k, nb, ynb_full):
    """dsh_lu_solve_squared_norm on banded factors of a small ensemble runs ONE launch (k_lu_band_solve_team<.., EPI>: the loaders of the backward sweep form the
    norm's terms, one lane per system adds them in index order at the end): the solution must equal dsh_lu_solve's and the norm Vector::squared_norm's sequential
    sum, bit for bit.  n = 900 (the squares do not fit LDS), K = 3 and the general banded factors (k = 10, 21: one wavefront per system) take the two-launch fallback: same bits."""
    rng = np.random.default_rng(7 * n + 13 * k + nb)
    c = H.HipContext(nbatch=nb)
    a = _banded(rng, nb, n, k, k, False)
    a[:, np.arange(n), np.arange(n)] += 0.5  # real interchanges, no near-singular pivots
    b = rng.standard_normal((nb, n))
    y = rng.standard_normal((nb, n)) if ynb_full else rng.standard_normal((1, n))
    atol = np.abs(rng.standard_normal(n)) * 1e-3 + 1e-6
    rtol = 1e-4
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    assert lu.band_width() == k
    x_sep = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x_sep)
    xs = np.asarray(x_sep.clone_as_vec()).reshape(nb, n)
    assert np.array_equal(xs, O.lu_solve(a, b)[0])
    x = H.HipVec.from_vec(b, c)
    yv = H.HipVec.from_vec(y, c if ynb_full else c.clone_with_nbatch(1))
    av = H.HipVec.from_vec(atol[None, :], c.clone_with_nbatch(1))
    out = C.c_double(0.0)
    L = c._L
    rc = L.dsh_lu_solve_squared_norm(lu._h, x.ptr, yv.ptr, yv.nb, av.ptr, 1, rtol, C.byref(out))
    assert rc == 0, L.dsh_last_error()
    assert np.array_equal(np.asarray(x.clone_as_vec()).reshape(nb, n), xs)
    term = xs / (np.abs(y) * rtol + atol[None, :])
    sq = term * term
    seq = np.cumsum(sq, axis=1)[:, -1] / float(n)  # cumsum adds in index order, like Vector::squared_norm (nalgebra_serial.rs:395-408)
    assert out.value == seq.max()
    assert out.value == x_sep.squared_norm(yv, av, rtol)  # and the stand-alone norm kernel's value
