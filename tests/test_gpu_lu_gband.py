"""GPU parity tests of the general-bandwidth batched banded LU (csrc/dsh_lu_gband.hpp; SURVEY 8(f) row 4, the `Sunmatrix_Band` analogue of
book/src/benchmarks/sundials.md:27-28): one wavefront per system, LAPACK dgbtrf-shaped partial pivoting with kl fill-in diagonals, half-bandwidths 5 .. 64.
The bar is the one of the K <= 4 register kernels: solutions BIT-IDENTICAL to the dense partial-pivoting LU (the dense kernels and the oracle's restatement of
nalgebra's LU), with and without real row interchanges, for operands in dense containers (probed or declared band) and in band containers; singular members are
reported like the dense kernels report them."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_lu_models import _banded, _pack_band

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


CASES = [(64, 5, 5), (100, 10, 10), (200, 20, 20), (128, 7, 3), (129, 3, 9), (130, 12, 0), (131, 0, 11), (257, 31, 32), (300, 33, 20), (400, 64, 64), (512, 40, 50),
         (513, 6, 6), (1024, 10, 10), (96, 16, 15), (640, 64, 1), (700, 1, 64)]


@pytest.mark.parametrize("n,kl,ku", CASES)
@pytest.mark.parametrize("dominant", [True, False])
def test_general_band_in_dense_containers_has_the_bits_of_the_dense_lu(H, O, n, kl, ku, dominant):
    """dsh_lu_factor probes (kl, ku) and the one-wavefront-per-system kernels eliminate the band only; the solution equals the oracle's dense LU and the dense
    kernels' bit for bit.  Ragged ensembles (the last workgroup partly idle), several right-hand sides, the handle switching between structures."""
    nb = 37 if n <= 300 else (11 if n <= 600 else 5)
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(100000 * n + 100 * kl + ku)
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
    # a second and a third right-hand side with the same factors
    for _ in range(2):
        b2 = rng.standard_normal((nb, n))
        x2 = H.HipVec.from_vec(b2, c)
        lu.solve_in_place(x2)
        assert np.array_equal(x2.clone_as_vec(), O.lu_solve(a, b2)[0])
    if n <= 300:  # the dense kernels on the same operand (DSH_LU_EXACT=1 in this tier): the same bits
        dense = H.HipLU(c, n)
        dense.set_structure(True)
        dense.factor(A)
        assert dense.band_width() == 0
        xd = H.HipVec.from_vec(b, c)
        dense.solve_in_place(xd)
        assert np.array_equal(xd.clone_as_vec(), xo)
    with pytest.raises(H.DiffsolHipError):
        lu.factors()  # banded factors have no dense image
    # the handle goes back and forth between structures: a narrow band (register kernels), then a full matrix (dense kernels), then the wide band again
    narrow = _banded(rng, nb, n, 1, 2, True)
    lu.factor(H.HipMat.from_array(narrow, c))
    assert lu.band_width() == 2
    x3 = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x3)
    assert np.array_equal(x3.clone_as_vec(), O.lu_solve(narrow, b)[0])
    if n <= 200:
        full = rng.standard_normal((nb, n, n))
        lu.factor(H.HipMat.from_array(full, c))
        assert lu.band_width() == 0
        x4 = H.HipVec.from_vec(b, c)
        lu.solve_in_place(x4)
        assert np.array_equal(x4.clone_as_vec(), O.lu_solve(full, b)[0])
    lu.factor(A)
    assert lu.band_width() == max(kl, ku)
    x5 = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x5)
    assert np.array_equal(x5.clone_as_vec(), xo)


@pytest.mark.parametrize("seed", range(6))
def test_random_bandwidths_up_to_64_against_the_oracle(H, O, seed):
    """random (n, kl, ku <= 64), random ensemble sizes (1 .. 70), random scaling of rows (pivoting on every step)"""
    rng = np.random.default_rng(4242 + seed)
    for _ in range(4):
        kl, ku = int(rng.integers(0, 65)), int(rng.integers(0, 65))
        if max(kl, ku) < 5:
            ku = 5 + int(rng.integers(0, 60))
        n = int(rng.integers(2 * (2 * kl + ku + 1), 2 * (2 * kl + ku + 1) + 200))
        n = min(n, 1024)
        if (2 * kl + ku + 1) * 2 > n:
            continue
        nb = int(rng.integers(1, 71)) if n <= 400 else int(rng.integers(1, 9))
        c = H.HipContext(nbatch=nb)
        a = _banded(rng, nb, n, kl, ku, False) * np.exp(rng.uniform(-3, 3, (nb, n, 1)))
        b = rng.standard_normal((nb, n))
        lu = H.HipLU(c, n)
        lu.factor(H.HipMat.from_array(a, c))
        assert lu.band_width() == max(kl, ku), (n, kl, ku)
        x = H.HipVec.from_vec(b, c)
        lu.solve_in_place(x)
        xo, _, _, rc = O.lu_solve(a, b)
        assert rc == 0 and np.array_equal(x.clone_as_vec(), xo), (n, kl, ku, nb)


def test_general_band_reports_singular_systems_like_the_dense_kernels(H, O):
    nb, n, kl, ku = 41, 120, 9, 7
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(5)
    a = _banded(rng, nb, n, kl, ku, True)
    a[7, :, 31] = 0.0   # a zero column: exact zero pivot at step 31 of system 7
    a[21, :, 0] = 0.0
    a[40, :, n - 1] = 0.0
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    assert lu.band_width() == 9 and lu.n_singular() == 3
    x = H.HipVec.from_vec(np.ones((nb, n)), c)
    with pytest.raises(H.DiffsolHipError) as e:
        lu.solve_in_place(x)
    assert "LuSolveFailed" in str(e.value) and "3 system" in str(e.value)
    ok = [b for b in range(nb) if b not in (7, 21, 40)]
    xo = O.lu_solve(a[ok], np.ones((len(ok), n)))[0]
    assert np.array_equal(x.clone_as_vec()[ok], xo)
    # the next factorisation of a regular ensemble starts a new count
    a2 = _banded(rng, nb, n, kl, ku, True)
    lu.factor(H.HipMat.from_array(a2, c))
    assert lu.n_singular() == 0


@pytest.mark.parametrize("n,kl,ku", [(100, 10, 10), (200, 20, 20), (150, 6, 30), (400, 64, 64), (90, 8, 5)])
@pytest.mark.parametrize("dominant", [True, False])
def test_general_band_containers_and_declared_bands_have_the_bits_of_the_dense_route(H, O, n, kl, ku, dominant):
    """band containers ((kl + ku + 1) n entries per member; dsh_lu_create_banded with k up to 64, dsh_lu_factor_packed) and dense containers with a DECLARED band
    (dsh_lu_factor_banded: no probe pass; a declaration wider than the content is an upper bound and must not change a bit)"""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb = 19
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(31 * n + kl)
    a = _banded(rng, nb, n, kl, ku, dominant)
    b = rng.standard_normal((nb, n))
    xo, _, _, rc = O.lu_solve(a, b)
    assert rc == 0
    Ab = H.HipVec.from_vec(_pack_band(a, kl, ku), c)
    h = C.c_void_p()
    k = max(kl, ku)
    assert L.dsh_lu_create_banded(c._h, n, nb, k, C.byref(h)) == 0
    try:
        assert L.dsh_lu_factor_packed(h, Ab.ptr, kl, ku) == 0 and L.dsh_lu_band_width(h) == k
        x = H.HipVec.from_vec(b, c)
        assert L.dsh_lu_solve(h, x.ptr) == 0
        assert np.array_equal(x.clone_as_vec(), xo)
        assert L.dsh_lu_factor(h, H.HipMat.from_array(a, c).ptr) != 0  # no room for dense factors in this handle
    finally:
        L.dsh_lu_destroy(h)
    lu = H.HipLU(c, n)  # the same container through an ordinary handle
    assert L.dsh_lu_factor_packed(lu._h, Ab.ptr, kl, ku) == 0
    x2 = H.HipVec.from_vec(b, c)
    lu.solve_in_place(x2)
    assert np.array_equal(x2.clone_as_vec(), xo)
    A = H.HipMat.from_array(a, c)
    for dkl, dku in ((kl, ku), (min(64, kl + 3), min(64, ku + 2))):  # declared exactly, declared wider
        if (2 * dkl + dku + 1) * 2 > n:
            continue
        lu2 = H.HipLU(c, n)
        assert L.dsh_lu_factor_banded(lu2._h, A.ptr, dkl, dku) == 0 and L.dsh_lu_band_width(lu2._h) == max(dkl, dku)
        x3 = H.HipVec.from_vec(b, c)
        lu2.solve_in_place(x3)
        assert np.array_equal(x3.clone_as_vec(), xo)


def test_general_band_multi_rhs_solve(H, O):
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb, n, kl, ku, nrhs = 9, 140, 11, 13, 4
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(77)
    a = _banded(rng, nb, n, kl, ku, False)
    lu = H.HipLU(c, n)
    lu.factor(H.HipMat.from_array(a, c))
    assert lu.band_width() == 13
    rhs = rng.standard_normal((nb, n, nrhs))
    B = H.HipMat.from_array(rhs, c)
    assert L.dsh_lu_solve_multi(lu._h, B.ptr, nrhs) == 0
    got = B.to_array()
    for r in range(nrhs):
        assert np.array_equal(got[:, :, r], O.lu_solve(a, rhs[:, :, r])[0])
