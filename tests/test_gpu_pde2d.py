"""GPU parity tests of the reference's 2-D PDE test models: heat2d (test_models/heat2d.rs: m x m grid, n = m^2, boundary
rows algebraic, half-bandwidth m) and foodweb (test_models/foodweb.rs: n = 2 nx^2, predators algebraic, consistent initialisation, half-bandwidth 2 nx) through the
host-driven BDF over the trait operations — M - cJ assembled on the declared band, factored by the general banded LU (csrc/dsh_lu_gband.hpp) — against
  * the CPU oracle (dense partial-pivot LU): every interpolated output bit for bit, all solver counters equal;
  * the reference's own solution tables (heat2d.rs:267-287, foodweb.rs:988-1050; acceptance norm < 20) and the ten OdeSolverStatistics counters of its insta
    snapshots (bdf.rs:2424-2490), which the oracle reproduces (tests/test_oracle_golden.py);
  * the dense LU route on the same problem: the same bits."""
import ctypes as C

import numpy as np
import pytest

from helpers import METHOD, ORACLE_MODEL, foodweb_out, heat2d_out, weighted_error_norm
from test_oracle_golden import SOLVER_COUNTERS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


def _problem(kats, which):
    tab = kats[which + "_table"]
    size = tab["mgrid"] if which == "heat2d" else tab["nx"]
    p = [1.0] if which == "heat2d" else [50.0, 1000.0]
    kw = dict(model_size=size, rtol=tab["problem_rtol"], atol=tab["problem_atol"], h0=1.0, method=METHOD["bdf"])
    return tab, size, p, kw


@pytest.mark.parametrize("which,size,np_", [("heat2d", 10, 1), ("heat2d", 7, 1), ("foodweb", 10, 2), ("foodweb", 6, 2)])
def test_init_mass_and_band_jacobian_of_the_2d_pde_models_match_the_oracle(H, O, which, size, np_):
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    nb = 13
    c = H.HipContext(nbatch=nb)
    mid, oid = H.MODELS[which], ORACLE_MODEL[which]
    n64, hm = C.c_int64(), C.c_int()
    assert L.dsh_model_info(mid, size, C.byref(n64), None, C.byref(hm), None) == 0 and hm.value == 1
    n = n64.value
    assert n == (size * size if which == "heat2d" else 2 * size * size)
    rng = np.random.default_rng(size)
    p = rng.uniform(0.5, 2.0, (nb, np_)) * ([1.0] if which == "heat2d" else [50.0, 1000.0])
    P = H.HipVec.from_vec(p, c)
    Y0 = H.HipVec.zeros(n, c)
    assert L.dsh_model_init(c._h, mid, size, nb, 0.0, P.ptr, Y0.ptr) == 0
    assert np.array_equal(Y0.clone_as_vec(), np.stack([O.model_init(oid, p[b], model_size=size) for b in range(nb)]))
    x, y = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
    X, Y = H.HipVec.from_vec(x, c), H.HipVec.from_vec(y, c)
    assert L.dsh_model_mass_gemv(c._h, mid, size, nb, 0.3, X.ptr, P.ptr, -0.7, Y.ptr) == 0
    assert np.array_equal(Y.clone_as_vec(), np.stack([O.model_mass_gemv(oid, x[b], p[b], y[b], -0.7, 0.3, size) for b in range(nb)]))
    M = H.HipMat.zeros(n, n, c)
    assert L.dsh_model_mass_matrix(c._h, mid, size, nb, 0.0, P.ptr, M.ptr) == 0
    mref = np.stack([np.stack([O.model_mass_gemv(oid, np.eye(n)[j], p[b], np.zeros(n), 0.0, 0.0, size) for j in range(n)], axis=1) for b in range(nb)])
    assert np.array_equal(M.to_array(), mref) and np.count_nonzero(mref[0]) == ((size - 2) ** 2 if which == "heat2d" else size * size)
    # the declared band, and the band-only evaluation of the Jacobian into a dense container
    jl, ju, ml, mu = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert L.dsh_model_band(mid, size, C.byref(jl), C.byref(ju), C.byref(ml), C.byref(mu)) == 0
    k = size if which == "heat2d" else 2 * size
    assert (jl.value, ju.value, ml.value, mu.value) == (k, k, 0, 0)
    xs = np.abs(x) + 0.1
    Xs = H.HipVec.from_vec(xs, c)
    J, Jb = H.HipMat.zeros(n, n, c), H.HipMat.zeros(n, n, c)
    assert L.dsh_model_jacobian(c._h, mid, size, nb, 0.1, Xs.ptr, P.ptr, J.ptr) == 0
    assert L.dsh_model_jacobian_band(c._h, mid, size, nb, 0.1, Xs.ptr, P.ptr, k, k, Jb.ptr) == 0
    ja = J.to_array()
    assert np.array_equal(ja, Jb.to_array())
    i, j = np.nonzero(ja[0])
    assert np.abs(i - j).max() == k


@pytest.mark.parametrize("which,snap", [("heat2d", "test_bdf_faer_sparse_heat2d"), ("foodweb", "test_bdf_faer_sparse_foodweb")])
def test_the_references_2d_pde_problems_through_bdf_and_the_general_banded_lu(H, O, kats, monkeypatch, which, snap):
    """bdf.rs:2424-2490 on the GPU (nbatch = 1, host-driven BDF over the trait operations): outputs equal the oracle's bit for bit, all 13 counters equal the oracle's,
    the ten solver counters equal the reference's insta snapshot, the model's out at the table's times is within the reference's acceptance norm — and the linear
    algebra really is the general banded LU."""
    tab, size, p, kw = _problem(kats, which)
    t = [pt["t"] for pt in tab["points"]]
    s = H.Solver(which, p, fused=False, **kw)
    y0 = s.state()["y"][0].copy()
    y, _ = s.solve_to_points(t[1:])
    o = O.OracleSolver(ORACLE_MODEL[which], p, **kw)
    yo0 = o.state()["y"][0].copy()
    yo, _ = o.solve_to_points(t[1:])
    assert np.array_equal(y0, yo0)  # foodweb: the consistent initialisation of the predators (state.rs:84-162) on the device
    assert np.array_equal(y, yo)
    assert s.stats() == o.stats()
    expected = kats["pde2d_snapshots"][snap]
    assert {k: s.stats()[k] for k in SOLVER_COUNTERS} == {k: expected[k] for k in SOLVER_COUNTERS}
    ys = np.concatenate([y0[None], y[:, 0]], axis=0)
    for k, pt in enumerate(tab["points"]):
        out = heat2d_out(ys[k], size)[None] if which == "heat2d" else foodweb_out(ys[k], size)
        assert weighted_error_norm(out, pt["y"], tab["atol"], tab["rtol"]) < 20.0, pt
    # the dense route on the same problem (DSH_LU_GBAND=0 is read when the library starts: a subprocess would be needed; DSH_LU_STRUCTURE is read per handle)
    monkeypatch.setenv("DSH_LU_STRUCTURE", "dense")
    sd = H.Solver(which, p, fused=False, **kw)
    yd, _ = sd.solve_to_points(t[1:])
    assert np.array_equal(yd, yo) and sd.stats() == o.stats()


@pytest.mark.parametrize("which,size", [("heat2d", 10), ("foodweb", 8), ("heat2d", 14)])
def test_lockstep_ensembles_of_the_2d_pde_models_are_bit_identical_to_the_oracle(H, O, which, size):
    """an ensemble with distinct members (diffusion scale of heat2d, growth-rate field parameters of foodweb) in lock-step: every member's output and the counters
    equal the oracle's batched run — the general banded LU on 29 different matrices per factorisation"""
    nb = 29
    rng = np.random.default_rng(size)
    p = rng.uniform(0.6, 1.6, (nb, 1)) if which == "heat2d" else rng.uniform(0.8, 1.2, (nb, 2)) * [50.0, 1000.0]
    kw = dict(nbatch=nb, model_size=size, rtol=1e-6, atol=[1e-6], h0=1.0, method=METHOD["bdf"])
    t = [0.01, 0.02, 0.08] if which == "heat2d" else [0.001, 0.01, 0.05]
    s = H.Solver(which, p, fused=False, **kw)
    o = O.OracleSolver(ORACLE_MODEL[which], p, **kw)
    y, _ = s.solve_to_points(t)
    yo, _ = o.solve_to_points(t)
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    st, so = s.state(), o.state()
    assert st["t"] == so["t"] and st["h"] == so["h"] and st["order"] == so["order"] and np.array_equal(st["y"], so["y"]) and np.array_equal(st["dy"], so["dy"])


def test_heat2d_with_tr_bdf2_and_esdirk34(H, O):
    """the SDIRK integrators on the banded DAE (sdirk.rs over the same LinearSolver): bit for bit against the oracle"""
    nb = 5
    p = np.linspace(0.7, 1.3, nb)[:, None]
    for method in ("tr_bdf2", "esdirk34"):
        kw = dict(nbatch=nb, model_size=10, rtol=1e-5, atol=[1e-5], h0=1.0, method=METHOD[method])
        s = H.Solver("heat2d", p, fused=False, **kw)
        o = O.OracleSolver(ORACLE_MODEL["heat2d"], p, **kw)
        y, _ = s.solve_to_points([0.01, 0.05])
        yo, _ = o.solve_to_points([0.01, 0.05])
        assert np.array_equal(y, yo) and s.stats() == o.stats()


def test_the_references_diffsl_form_of_heat2d_on_the_device(H, O, kats):
    """test_bdf_faer_sparse_heat2d_diffsl on the GPU: the reference's generated DiffSL text (sparse D_ij, Mass_ij, init_i; tests/diffsl_models.py::heat2d) compiled by
    the product's front end and hiprtc, integrated by the host-driven BDF with the front end's structural band (10, 10) declared — M - cJ assembled on the band and
    factored by the general banded LU.  Outputs and all counters equal the oracle's run of the SAME generated model (its host twin) bit for bit; the closure
    snapshot's ten solver counters; the reference's table; an ensemble of 21 members with distinct diffusion scales bitwise."""
    import diffsl_models as DM
    from diffsol_amd import diffsl as fe
    code = DM.heat2d(10)
    model = fe.DiffslModel(code)
    assert model.n == 100 and model.has_mass and tuple(model.band) == (10, 10, 0, 0)
    tab = kats["heat2d_table"]
    kw = dict(rtol=tab["problem_rtol"], atol=tab["problem_atol"], h0=1.0, method=METHOD["bdf"])
    t = [pt["t"] for pt in tab["points"]]
    s = H.Solver(model, [[1.0]], **kw)
    y, _ = s.solve_to_points(t[1:])
    mid = DM.host_model(O, code)
    o = O.OracleSolver(mid, [1.0], **kw)
    yo, _ = o.solve_to_points(t[1:])
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    expected = kats["pde2d_snapshots"]["test_bdf_faer_sparse_heat2d"]
    assert {k: s.stats()[k] for k in SOLVER_COUNTERS} == {k: expected[k] for k in SOLVER_COUNTERS}
    for k, pt in enumerate(tab["points"][1:]):
        assert weighted_error_norm(heat2d_out(y[k, 0], 10)[None], pt["y"], tab["atol"], tab["rtol"]) < 20.0, pt
    nb = 21
    p = np.linspace(0.7, 1.4, nb)[:, None]
    kw2 = dict(nbatch=nb, rtol=1e-6, atol=[1e-6], h0=1.0, method=METHOD["bdf"])
    s2 = H.Solver(fe.DiffslModel(code), p, **kw2)
    o2 = O.OracleSolver(mid, p, **kw2)
    y2, _ = s2.solve_to_points([0.01, 0.05])
    yo2, _ = o2.solve_to_points([0.01, 0.05])
    assert np.array_equal(y2, yo2) and s2.stats() == o2.stats()


def test_the_references_diffsl_form_of_foodweb_on_the_device(H, O, kats):
    """test_bdf_faer_sparse_foodweb_diffsl on the GPU: the reference's generated DiffSL text (tests/diffsl_models.py::foodweb — separate species blocks, sparse diffusion
    operator, sin / pow tensors, algebraic predators, a dense-band Jacobian of half-bandwidth 100: the dense LU) compiled by the front end and hiprtc; the consistent
    initialisation and BDF on the device equal the oracle's run of the generated host twin bit for bit, counters included, and the model's out_i meets the reference's table."""
    import diffsl_models as DM
    from diffsol_amd import diffsl as fe
    code = DM.foodweb(10)
    model = fe.DiffslModel(code)
    assert model.n == 200 and model.has_mass
    tab = kats["foodweb_table"]
    kw = dict(rtol=tab["problem_rtol"], atol=tab["problem_atol"], h0=1.0, method=METHOD["bdf"])
    t = [pt["t"] for pt in tab["points"]]
    s = H.Solver(model, [[0.0]], **kw)
    y0 = s.state()["y"][0].copy()
    y, _ = s.solve_to_points(t[1:])
    mid = DM.host_model(O, code)
    o = O.OracleSolver(mid, [0.0], **kw)
    yo0 = o.state()["y"][0].copy()
    yo, _ = o.solve_to_points(t[1:])
    assert np.array_equal(y0, yo0) and np.array_equal(y, yo) and s.stats() == o.stats()
    ys = np.concatenate([y0[None], y[:, 0]], axis=0)
    for k, pt in enumerate(tab["points"]):
        assert weighted_error_norm(O.model_out(mid, ys[k], [0.0]), pt["y"], tab["atol"], tab["rtol"]) < 20.0, pt


def test_the_references_tr_bdf2_heat2d_problem_on_the_device(H, O, kats):
    """test_tr_bdf2_faer_sparse_heat2d (sdirk.rs:995-1000) on the GPU: TR-BDF2 over the trait operations with the general banded LU, the table's problem tolerances —
    bitwise against the oracle, the table within the reference's norm"""
    tab = kats["heat2d_table"]
    kw = dict(model_size=10, rtol=tab["problem_rtol"], atol=tab["problem_atol"], h0=1.0, method=METHOD["tr_bdf2"])
    t = [pt["t"] for pt in tab["points"]]
    s = H.Solver("heat2d", [1.0], fused=False, **kw)
    o = O.OracleSolver(ORACLE_MODEL["heat2d"], [1.0], **kw)
    y, _ = s.solve_to_points(t[1:])
    yo, _ = o.solve_to_points(t[1:])
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    for k, pt in enumerate(tab["points"][1:]):
        assert weighted_error_norm(heat2d_out(y[k, 0], 10)[None], pt["y"], tab["atol"], tab["rtol"]) < 20.0, pt
