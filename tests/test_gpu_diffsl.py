"""DiffSL models on the GPU: DiffSL text -> front end -> hiprtc -> the library's own kernel templates instantiated for the user's model.

Oracle: the SAME DiffSL text through the front end's host target, compiled with g++ and loaded into the CPU restatement as an external model
(tests/diffsl_models.py host_model).  Both sides evaluate the generated expressions in the same order with the same deterministic elementary functions,
so every comparison below is bit for bit.  The front end itself is checked independently on the CPU (tests/test_diffsl_front.py: generated model ==
hand-written model)."""
import ctypes as C

import numpy as np
import pytest

import diffsl_models as D
from helpers import METHOD, robertson_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


@pytest.fixture(scope="module")
def fe():
    from diffsol_amd import diffsl
    return diffsl


@pytest.fixture
def det_pow(O):
    O.set_det_pow(True)
    yield
    O.set_det_pow(False)


CASES = {"robertson": D.ROBERTSON_DAE, "rlc": D.RLC, "logistic": D.LOGISTIC, "zoo": D.ZOO, "heat16": D.heat1d(16), "spm": D.spm(20), "heat_dae": D.HEAT_DAE}


def _sample(name, dims, rng, nb):
    n = dims["n"]
    if name == "spm":
        x = np.concatenate([rng.uniform(0, 1, (nb, 2)), rng.uniform(2e3, 2e4, (nb, 20)), rng.uniform(2e4, 4.5e4, (nb, 20))], axis=1)
    else:
        x = rng.uniform(0.2, 1.5, (nb, n))
    return x, rng.standard_normal((nb, n)), rng.uniform(0.5, 2.0, (nb, dims["nparams"]))


@pytest.mark.parametrize("name", list(CASES))
def test_every_operator_of_a_compiled_model_matches_the_cpu_model_bitwise(H, O, fe, name):
    _check_every_operator(H, O, fe, name)


@pytest.mark.parametrize("name", ["heat16", "spm", "heat_dae"])
def test_outlined_form_of_large_models_gives_the_same_bits(H, O, fe, name, monkeypatch):
    """Models whose inline device source would exceed 4 MB (the reference's pybamm_dfn.diffsl: 962 states, 12 MB) are emitted OUTLINED — one __noinline__
    function per component behind a dispatching switch, instantiated once for all kernels through the concrete accessor types of dsh_jit_dyn_kernels.hpp —
    because one function of that size does not get through the device compiler.  Forced here on the run-time-sized test models: every operator (right-hand
    side, J v, dense Jacobian, initial state, mass product and matrix, roots, outputs) must still equal the host model bit for bit."""
    monkeypatch.setenv("DSH_DIFFSL_OUTLINE", "1")
    src = fe.generate(CASES[name], fe.TARGET_HIP_DYNAMIC)[0]
    assert "DSH_JIT_OUTLINED" in src and src.count("__attribute__((noinline))") >= 3 * O.model_dims(D.host_model(O, CASES[name]))["n"]
    monkeypatch.setenv("DSH_DIFFSL_OUTLINE", "0")
    assert "DSH_JIT_OUTLINED" not in fe.generate(CASES[name], fe.TARGET_HIP_DYNAMIC)[0]
    monkeypatch.setenv("DSH_DIFFSL_OUTLINE", "1")
    _check_every_operator(H, O, fe, name)


def _check_every_operator(H, O, fe, name):
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    m = fe.DiffslModel(CASES[name])
    mid = D.host_model(O, CASES[name])
    dims = O.model_dims(mid)
    assert (m.n, m.nparams, m.nroots, m.nout, m.has_mass) == (dims["n"], dims["nparams"], dims["nroots"], dims["nout"], dims["has_mass"])
    assert m.form == (fe.FORM_STATIC if m.n <= 8 and m.nroots <= 1 else fe.FORM_DYNAMIC)
    nb, n, t = 70, m.n, 0.37
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(n)
    x, v, p = _sample(name, dims, rng, nb)
    X, V, P, Y = H.HipVec.from_vec(x, c), H.HipVec.from_vec(v, c), H.HipVec.from_vec(p, c), H.HipVec.zeros(n, c)
    i64, i32 = C.c_int64(), C.c_int()
    assert L.dsh_model_info(m.model_id, 0, C.byref(i64), None, C.byref(i32), None) == 0 and i64.value == n and bool(i32.value) == m.has_mass
    assert L.dsh_model_rhs(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, Y.ptr) == 0
    assert np.array_equal(Y.clone_as_vec(), np.stack([O.model_rhs(mid, x[b], p[b], t) for b in range(nb)]))
    assert L.dsh_model_jac_mul(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, V.ptr, Y.ptr) == 0
    assert np.array_equal(Y.clone_as_vec(), np.stack([O.model_jac_mul(mid, x[b], p[b], v[b], t) for b in range(nb)]))
    J = H.HipMat.zeros(n, n, c)
    assert L.dsh_model_jacobian(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, J.ptr) == 0
    jref = np.empty((nb, n, n))
    for b in range(nb):
        for j in range(n):
            e = np.zeros(n); e[j] = 1.0
            jref[b, :, j] = O.model_jac_mul(mid, x[b], p[b], e, t)
    assert np.array_equal(J.to_array(), jref)
    assert L.dsh_model_init(c._h, m.model_id, 0, nb, 0.0, P.ptr, Y.ptr) == 0
    assert np.array_equal(Y.clone_as_vec(), np.stack([O.model_init(mid, p[b]) for b in range(nb)]))
    y0 = rng.standard_normal((nb, n))
    Y = H.HipVec.from_vec(y0, c)
    assert L.dsh_model_mass_gemv(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, 0.6, Y.ptr) == 0
    assert np.array_equal(Y.clone_as_vec(), np.stack([O.model_mass_gemv(mid, x[b], p[b], y0[b], 0.6, t) for b in range(nb)]))
    if m.has_mass:
        Mm = H.HipMat.zeros(n, n, c)
        assert L.dsh_model_mass_matrix(c._h, m.model_id, 0, nb, t, P.ptr, Mm.ptr) == 0
        mref = np.stack([np.stack([O.model_mass_gemv(mid, np.eye(n)[j], p[b], np.zeros(n), 0.0, t) for j in range(n)], axis=1) for b in range(nb)])
        assert np.array_equal(Mm.to_array(), mref)
    if m.nroots:
        G = H.HipVec.zeros(m.nroots, c)
        assert L.dsh_model_root(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, G.ptr) == 0
        assert np.array_equal(G.clone_as_vec(), np.stack([O.model_root(mid, x[b], p[b], t, 0, max_roots=8) for b in range(nb)]))
    if m.nout:
        G = H.HipVec.zeros(m.nout, c)
        assert L.dsh_model_out(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, G.ptr) == 0
        assert np.array_equal(G.clone_as_vec(), np.stack([O.model_out(mid, x[b], p[b], t) for b in range(nb)]))
    else:
        assert L.dsh_model_out(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, Y.ptr) < 0
    old_id = m.model_id
    m.release()
    assert L.dsh_model_rhs(c._h, old_id, 0, nb, t, X.ptr, P.ptr, Y.ptr) < 0  # released id: loud error, not a stale kernel


def _pair(H, O, fe, code, p, method, **tol):
    m = fe.DiffslModel(code)
    mid = D.host_model(O, code)
    nb = len(p)
    s = H.Solver(m, p, nbatch=nb, method=METHOD[method], **tol)
    o = O.OracleSolver(mid, p, nbatch=nb, method=METHOD[method], **tol)
    return m, s, o


def test_lockstep_bdf_on_a_compiled_robertson_uses_the_fused_kernels_and_matches_the_oracle_bitwise(H, O, fe):
    p = robertson_params(64, seed=3)
    m, s, o = _pair(H, O, fe, D.ROBERTSON_ODE, p, "bdf", rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
    assert s.fused and m.form == fe.FORM_STATIC
    times = [0.4, 4.0, 40.0, 400.0, 4000.0]
    y, _ = s.solve_to_points(times)
    yo, _ = o.solve_to_points(times)
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    # and the built-in model gives the same trajectories to rounding (its hand-written J v orders one product differently)
    yb, _ = H.Solver("robertson_ode", p, nbatch=64, model_size=1, rtol=1e-4, atol=[1e-8, 1e-14, 1e-6]).solve_to_points(times)
    assert np.allclose(y, yb, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
def test_lockstep_dae_with_mass_matrix_and_consistent_initialisation(H, O, fe, method):
    p = robertson_params(16, seed=4)
    m, s, o = _pair(H, O, fe, D.ROBERTSON_DAE, p, method, rtol=1e-4, atol=[1e-8, 1e-6, 1e-6])
    y, _ = s.solve_to_points([0.4, 4.0, 40.0])
    yo, _ = o.solve_to_points([0.4, 4.0, 40.0])
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    assert np.allclose(y.sum(axis=2), 1.0, atol=1e-5)


def test_lockstep_rlc_with_event_from_diffsl(H, O, fe):
    p = np.tile([100.0, 1.0, 1e-3, 10.0, 100.0, 0.01], (4, 1))
    m, s, o = _pair(H, O, fe, D.RLC, p, "esdirk34", rtol=1e-6, atol=[1e-6] * 4)
    _, _, reason = s.solve(1.0)
    o.solve(1.0)
    assert reason == 1 and s.root_info() == o.root_info() and s.stats() == o.stats()
    st, t_root = s.state(), s.root_info()[0]  # the solver moved its state back to the root time (state_mut_back): y and dy from the step's interpolants
    assert st["t"] == t_root and np.array_equal(st["y"], o.interpolate(t_root)) and np.array_equal(st["dy"], o.interpolate_dy(t_root))


@pytest.mark.parametrize("name,method,times", [("heat16", "tr_bdf2", [0.01, 0.05]), ("heat_dae", "bdf", [0.005, 0.02]), ("spm", "bdf", [60.0, 600.0])])
def test_lockstep_run_time_sized_models_from_diffsl(H, O, fe, name, method, times):
    rng = np.random.default_rng(7)
    p = rng.uniform(0.6, 1.4, (9, 1))
    m, s, o = _pair(H, O, fe, CASES[name], p, method, rtol=1e-6, atol=[1e-6])
    assert m.form == fe.FORM_DYNAMIC and not s.fused
    y, _ = s.solve_to_points(times)
    yo, _ = o.solve_to_points(times)
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    if name == "heat_dae":
        assert np.all(np.abs(y[..., 0]) < 1e-12) and np.all(np.abs(y[..., -1]) < 1e-12) and np.all(y[..., 1:-1] > 0.0)


def test_spm_from_diffsl_stops_at_its_surface_concentration_limit_like_the_oracle(H, O, fe):
    p = np.full((3, 1), 1.2)
    m, s, o = _pair(H, O, fe, D.spm(20), p, "bdf", rtol=1e-6, atol=[1e-6])
    _, _, reason = s.solve(20000.0)
    o.solve(20000.0)
    assert reason == 1 and s.root_info() == o.root_info() and s.root_info()[0] < 20000.0


@pytest.mark.parametrize("group", [1, 64])
def test_device_resident_bdf_on_a_compiled_model_is_bit_identical_to_independent_cpu_solves(H, O, fe, det_pow, group):
    p = robertson_params(300, seed=8)
    tol = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
    m = fe.DiffslModel(D.ROBERTSON_ODE)
    mid = D.host_model(O, D.ROBERTSON_ODE)
    s = H.Solver(m, p, nbatch=len(p), **tol)
    t_eval = [0.4, 4.0, 40.0, 400.0, 4000.0, 40000.0]
    y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group)
    yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=group, method=0, **tol)
    assert failed == 0 and (mem["status"] == 0).all()
    assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)))


def test_device_resident_esdirk34_with_per_member_events_on_the_compiled_rlc(H, O, fe, det_pow):
    nb = 120
    rng = np.random.default_rng(5)
    R, Cc = rng.uniform(50.0, 200.0, nb), np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
    p = np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, 0.03)], axis=1)
    tol = dict(rtol=1e-6, atol=[1e-6] * 4)
    m = fe.DiffslModel(D.RLC)
    mid = D.host_model(O, D.RLC)
    s = H.Solver(m, p, nbatch=nb, method=METHOD["esdirk34"], **tol)
    t_eval = [0.002, 0.005, 0.01, 0.02, 0.05]
    y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=1, method=2, **tol)
    ref = O.solve_dense_independent.last_roots
    assert failed == 0 and (mem["status"] == 0).all() and 0 < (mem["root_idx"] >= 0).sum() < nb
    assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True)
    assert np.array_equal(mem["root_idx"], ref["root_idx"]) and np.array_equal(mem["ncols"], ref["ncols"]) and np.array_equal(mem["t_root"], ref["t_root"], equal_nan=True)


def test_wavefront_per_member_bdf_on_run_time_sized_diffsl_models_with_per_member_events(H, O, fe, det_pow):
    """n <= 64, identity mass: one wavefront per member, the kernel of BASELINE config 4 instantiated by hiprtc for the DiffSL model (its components
    behind a switch on the lane's component).  States, counters, event times bit-identical to independent CPU solves of the same model."""
    cur = np.linspace(0.6, 1.4, 24)[:, None]
    tol = dict(rtol=1e-6, atol=[1e-6])
    code = D.spm(20)
    m, mid = fe.DiffslModel(code, lane_resident=False), D.host_model(O, code)  # lane_resident=False: no banded twin, the wavefront-per-member kernel runs
    assert m.lane_model_id is None
    s = H.Solver(m, cur, nbatch=24, **tol)
    t_eval = [600.0, 3000.0, 9000.0, 15000.0]
    y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    yo, so, failed = O.solve_dense_independent(mid, cur, t_eval, nthreads=8, group=1, method=0, **tol)
    ref = O.solve_dense_independent.last_roots
    assert failed == 0 and (mem["status"] == 0).all() and (mem["root_idx"] >= 0).sum() > 3 and np.nanmax(mem["t_root"]) - np.nanmin(mem["t_root"]) > 100.0
    assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True)
    assert np.array_equal(mem["root_idx"], ref["root_idx"]) and np.array_equal(mem["ncols"], ref["ncols"]) and np.array_equal(mem["t_root"], ref["t_root"], equal_nan=True)
    code = D.heat1d(24)
    m, mid = fe.DiffslModel(code, lane_resident=False), D.host_model(O, code)
    p = np.random.default_rng(2).uniform(0.5, 2.0, (10, 1))
    s = H.Solver(m, p, nbatch=10, **tol)
    y, tot, mem = s.solve_dense_adaptive([0.01, 0.1], want_member_stats=True, group=1)
    yo, so, failed = O.solve_dense_independent(mid, p, [0.01, 0.1], nthreads=4, group=1, method=0, **tol)
    assert failed == 0 and np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)))
    # the SDIRK methods in the same form (k_sdirk_wave_member, instantiated by hiprtc for the model): the battery model through its events, TR-BDF2 and ESDIRK34
    code = D.spm(20)
    m, mid = fe.DiffslModel(code, lane_resident=False), D.host_model(O, code)
    for hm, om in ((H.METHOD_TR_BDF2, 1), (H.METHOD_ESDIRK34, 2)):
        s = H.Solver(m, cur, nbatch=24, method=hm, **tol)
        y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
        yo, so, failed = O.solve_dense_independent(mid, cur, t_eval, nthreads=8, group=1, method=om, **tol)
        ref = O.solve_dense_independent.last_roots
        assert failed == 0 and (mem["status"] == 0).all() and (mem["root_idx"] >= 0).sum() > 3
        assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True)
        assert np.array_equal(mem["root_idx"], ref["root_idx"]) and np.array_equal(mem["ncols"], ref["ncols"]) and np.array_equal(mem["t_root"], ref["t_root"], equal_nan=True)


DAE10 = """
in = [k]
k { 1.0 }
u_i { (0:8): x = 1, p = 0.7, q = 0.2 }
dudt_i { (0:8): dxdt = 0, dpdt = 0, dqdt = 0 }
M_i { dxdt_i, 0, 0 }
F_i { (-k * x_i) * (1.0 + p) + q, p * p + p - x_i[0:1], q - 0.1 * p * x_i[7:8] }
"""


def test_wavefront_per_member_integrators_take_daes_with_a_consistent_initialisation_on_the_device(H, O, fe, det_pow):
    """mass matrices in the wavefront-per-member kernel (run-time-sized DiffSL models, n <= 48).  The kernel makes the initial state
    consistent itself — InitOp's Newton iteration with the backtracking line search, one row per lane (state.rs:84-162, op/init.rs, line_search.rs:84-201) —
    and carries M in the residual M (y - y0 + psi) - c f and in M - cJ.  A heat equation with algebraic boundary unknowns, and a ten-state DAE whose two
    algebraic unknowns start OFF their nonlinear constraints (p^2 + p = x_0, q = 0.1 p x_7): states and counters of every member bit-identical to
    independent CPU solves of the host twin."""
    tol = dict(rtol=1e-6, atol=[1e-6])
    for code, p, t_eval in ((D.HEAT_DAE, np.random.default_rng(3).uniform(0.5, 2.0, (9, 1)), [0.005, 0.02, 0.1]),
                            (DAE10, np.linspace(0.5, 3.0, 11)[:, None], [0.1, 0.5, 2.0])):
        m, mid = fe.DiffslModel(code, lane_resident=False), D.host_model(O, code)
        assert m.has_mass and m.form == fe.FORM_DYNAMIC
        for hm, om in ((H.METHOD_BDF, 0), (H.METHOD_TR_BDF2, 1), (H.METHOD_ESDIRK34, 2)):  # k_bdf_wave_member, k_sdirk_wave_member<., 3>, <., 4>
            s = H.Solver(m, p, nbatch=len(p), method=hm, **tol)
            y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
            yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=4, group=1, method=om, **tol)
            assert failed == 0 and (mem["status"] == 0).all(), (hm, mem["status"])
            assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2))), hm
    assert abs(yo[0, 0, 8] ** 2 + yo[0, 0, 8] - yo[0, 0, 0]) < 1e-5  # on the constraint


def test_banded_models_get_the_lane_per_member_bdf_with_state_in_memory_and_a_banded_lu(H, O, fe, det_pow):
    """8 < n <= 64, identity mass, Jacobian bandwidth <= 4: the DiffSL model is compiled a second time in the lane-per-member form (BAND_K, jac_band) and
    per-member device-resident solves run on k_bdf_adaptive with the BDF state in per-lane memory and the banded LU in registers — the kernel of the
    n <= 4 models, not the wavefront-per-member one.  Same bits as independent CPU solves: states, counters, event times (terminal-voltage cut-off)."""
    tol = dict(rtol=1e-6, atol=[1e-6])
    saw_events = False
    for code, p, t_eval in [(D.spm(5, voltage=True), np.linspace(0.6, 1.4, 70)[:, None], [600.0, 3000.0, 9000.0, 20000.0]),
                            (D.heat1d(12), np.random.default_rng(2).uniform(0.5, 2.0, (70, 1)), [0.01, 0.1])]:
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        assert m.form == fe.FORM_DYNAMIC and m.lane_model_id is not None
        s = H.Solver(m, p, nbatch=len(p), **tol)
        y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
        yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=1, method=0, **tol)
        ref = O.solve_dense_independent.last_roots
        assert failed == 0 and (mem["status"] == 0).all()
        assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True)
        assert np.array_equal(mem["root_idx"], ref["root_idx"]) and np.array_equal(mem["ncols"], ref["ncols"]) and np.array_equal(mem["t_root"], ref["t_root"], equal_nan=True)
        # the host-driven lock-step path of the same model object is untouched by the twin
        O.set_det_pow(False)  # the host-side integrators call libm's pow like the reference; the deterministic pow belongs to the device-resident kernels
        yl, _ = H.Solver(m, p[:5], nbatch=5, **tol).solve_to_points(t_eval[:1])
        ol, _ = O.OracleSolver(mid, p[:5], nbatch=5, **tol).solve_to_points(t_eval[:1])
        O.set_det_pow(True)
        assert np.array_equal(yl, ol)
        if mem["root_idx"].max() >= 0:
            saw_events = True
    assert saw_events  # the battery members reach a cut-off voltage, each at its own time


def test_singular_mass_battery_dae_runs_on_the_lane_per_member_banded_bdf_with_the_initialisation_on_the_device(H, O, fe, det_pow):
    """a DIAGONAL mass matrix in k_bdf_lane_banded.  The battery model with the terminal voltage as an algebraic state (n = 13 and 43,
    M = diag(1.., 0, ..1), bandwidth 2; BASELINE configs[3] as worded): every lane makes its own member consistent (InitOp's Newton iteration with the line
    search on the banded LU, state.rs:84-162), then integrates with M in the residual M (y - y0 + psi) - c f and in M - cJ (op/bdf.rs:240-300).  States,
    counters and stop times (the voltage cut-off, now a condition on a STATE) of every member equal independent CPU solves bit for bit — per member and in
    lock-step groups of 64."""
    tol = dict(rtol=1e-6, atol=[1e-6])
    saw_events = False
    for m_shells, nb, group in ((5, 70, 1), (20, 70, 1), (20, 150, 64)):
        code = D.spm_dae(m_shells)
        p = np.linspace(0.6, 1.4, nb)[:, None]
        # a lock-step group stops only where ALL its members see the root (the reference's batched root finding, kRsRootBatchMismatch otherwise): it ends before the first cut-off
        t_eval = [600.0, 3000.0, 9000.0, 20000.0] if group == 1 else [600.0, 1500.0]
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        assert m.has_mass and m.lane_model_id is not None
        s = H.Solver(m, p, nbatch=nb, **tol)
        y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group)
        yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=group, method=0, **tol)
        ref = O.solve_dense_independent.last_roots
        assert failed == 0 and (mem["status"] == 0).all(), mem["status"]
        assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True)
        assert np.array_equal(mem["root_idx"], ref["root_idx"]) and np.array_equal(mem["ncols"], ref["ncols"]) and np.array_equal(mem["t_root"], ref["t_root"], equal_nan=True)
        saw_events = saw_events or mem["root_idx"].max() >= 0
        assert np.isfinite(y[0]).all() and (y[0][:, 2 + m_shells] > 3.105 - 1e-6).all() and (y[0][:, 2 + m_shells] < 4.1).all()  # the algebraic state: a voltage between its two stops (a member stopped before the first save point holds its event state there)
    assert saw_events


@pytest.mark.parametrize("method", ["tr_bdf2", "esdirk34"])
def test_banded_models_also_get_the_lane_per_member_sdirk_integrators(H, O, fe, det_pow, method):
    """TR-BDF2 / ESDIRK34 per member on the device for a banded run-time-sized model (k_sdirk_resident in its banded form), with stop conditions."""
    tol = dict(rtol=1e-6, atol=[1e-6])
    for code, p, t_eval in [(D.heat1d(12), np.random.default_rng(2).uniform(0.5, 2.0, (40, 1)), [0.01, 0.1]),
                            (D.spm(5, voltage=True), np.linspace(0.6, 1.4, 40)[:, None], [600.0, 3000.0, 9000.0, 20000.0])]:
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        s = H.Solver(m, p, nbatch=len(p), method=METHOD[method], **tol)
        y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
        yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=1, method=METHOD[method], **tol)
        ref = O.solve_dense_independent.last_roots
        assert failed == 0 and (mem["status"] == 0).all()
        assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True)
        assert np.array_equal(mem["root_idx"], ref["root_idx"]) and np.array_equal(mem["ncols"], ref["ncols"]) and np.array_equal(mem["t_root"], ref["t_root"], equal_nan=True)
    # wavefront lock-step groups (the reference's batched semantics with nbatch = 64) on the same banded kernels
    code = D.heat1d(12)
    m, mid = fe.DiffslModel(code), D.host_model(O, code)
    p = np.random.default_rng(4).uniform(0.5, 2.0, (150, 1))
    s = H.Solver(m, p, nbatch=150, method=METHOD[method], **tol)
    y, tot, mem = s.solve_dense_adaptive([0.01, 0.1], want_member_stats=True, group=64)
    yo, so, failed = O.solve_dense_independent(mid, p, [0.01, 0.1], nthreads=4, group=64, method=METHOD[method], **tol)
    assert failed == 0 and np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)))
    # the built-in heat model takes the same route (DynLane<heat1d, 20, ...>)
    p = np.random.default_rng(3).uniform(0.5, 2.0, (20, 1))
    s = H.Solver("heat1d", p, nbatch=20, model_size=20, method=METHOD[method], **tol)
    y, tot, mem = s.solve_dense_adaptive([0.01, 0.1], want_member_stats=True, group=1)
    yo, so, failed = O.solve_dense_independent(7, p, [0.01, 0.1], model_size=20, nthreads=4, group=1, method=METHOD[method], **tol)
    assert failed == 0 and np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)))


def test_random_models_give_the_same_bits_on_the_device_and_on_the_host(H, O, fe):
    """20 random three-state models (every operator, eight elementary functions, min / max): right-hand side, J v and the dense Jacobian of the hiprtc-compiled
    model equal the g++-compiled host model bit for bit — the two emitters and the deterministic elementary functions agree on arbitrary expressions."""
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    rng = np.random.default_rng(77)
    names = ["x", "y", "z", "a", "b", "t"]
    nb = 64
    c = H.HipContext(nbatch=nb)
    for k in range(20):
        exprs = [D.random_expr(rng, 4, names)[0] for _ in range(3)]
        code = "in = [a, b]\na { 1 } b { 1 }\nu_i { x = 0.4, y = 0.9, z = 1.7 }\nF_i {\n" + ",\n".join(exprs) + "\n}\n"
        m, mid = fe.DiffslModel(code), D.host_model(O, code, opt="-O1")
        x, v, p, t = rng.uniform(0.3, 2.0, (nb, 3)), rng.standard_normal((nb, 3)), rng.uniform(0.5, 1.5, (nb, 2)), 0.37
        X, V, P, Y = H.HipVec.from_vec(x, c), H.HipVec.from_vec(v, c), H.HipVec.from_vec(p, c), H.HipVec.zeros(3, c)
        assert L.dsh_model_rhs(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, Y.ptr) == 0
        assert np.array_equal(Y.clone_as_vec(), np.stack([O.model_rhs(mid, x[b], p[b], t) for b in range(nb)])), code
        assert L.dsh_model_jac_mul(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, V.ptr, Y.ptr) == 0
        assert np.array_equal(Y.clone_as_vec(), np.stack([O.model_jac_mul(mid, x[b], p[b], v[b], t) for b in range(nb)])), code
        m.release()


def test_a_model_that_does_not_compile_is_rejected_with_the_compiler_log(H, fe):
    from diffsol_amd import _ffi, DiffsolHipError
    L = _ffi.load_device_lib()
    mid = C.c_int()
    rc = L.dsh_model_compile(b"namespace dsh { struct JitModel { static constexpr int N = 2; this is not C++ }; }", 0, 2, 1, 0, 0, 0, C.byref(mid))
    assert rc < 0 and b"error" in L.dsh_last_error()
    good = fe.generate(D.LOGISTIC, fe.TARGET_HIP_STATIC)[0]
    rc = L.dsh_model_compile(good.encode(), 0, 3, 2, 1, 0, 0, C.byref(mid))  # wrong n for this source
    assert rc < 0 and b"dimensions do not match" in L.dsh_last_error()


def test_fast_forcing_terms_sin_cos_of_huge_arguments_are_the_same_bits_on_device_and_host_and_close_to_libm(H, O, fe):
    """a forcing term sin(w t) with w t > 1e6 (a 1 MHz source integrated for 1 s) used to return NaN.  The deterministic sin / cos now
    reduce large arguments with an integer Payne-Hanek step: device == host twin bit for bit, within 2 ulp of libm, over the whole double range."""
    code = """
    in = [w]
    w { 1.0 }
    u_i { x = 1.0, y = 0.0, z = 0.0 }
    F_i { sin(w * t) - x, cos(w * t) - y, tan(w * t) * 0.0 + sin(-w * t) - z }
    """
    from diffsol_amd import _ffi
    L = _ffi.load_device_lib()
    m = fe.DiffslModel(code)
    mid = D.host_model(O, code)
    rng = np.random.default_rng(8)
    w = np.concatenate([10.0 ** rng.uniform(6, 300, 120), [1e9, 1e22, 2.0 ** 1023, 1e6 + 0.5]])
    nb, t = len(w), 1.0
    c = H.HipContext(nbatch=nb)
    x = np.zeros((nb, 3))
    X, P, Y = H.HipVec.from_vec(x, c), H.HipVec.from_vec(w[:, None], c), H.HipVec.zeros(3, c)
    assert L.dsh_model_rhs(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, Y.ptr) == 0
    dev = Y.clone_as_vec()
    host = np.stack([O.model_rhs(mid, x[b], w[b:b + 1], t) for b in range(nb)])
    assert np.isfinite(dev).all() and np.array_equal(dev, host)
    assert np.max(np.abs(dev[:, 0] - np.sin(w)) / np.spacing(np.abs(np.sin(w)))) <= 3.0
    assert np.max(np.abs(dev[:, 1] - np.cos(w)) / np.spacing(np.abs(np.cos(w)))) <= 3.0
    assert np.array_equal(dev[:, 2], -dev[:, 0])  # odd symmetry (the tan term only checks that tan of a huge argument is finite)


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
def test_hybrid_diffsl_models_reset_at_every_event_on_the_device_like_the_oracle(H, O, fe, method):
    """reset_i on the HIP backend: the model's reset operator is compiled into both device forms (dsh_model_reset), solve_dense of a hybrid model runs host-driven
    (the device-resident kernels stop at an event), moves the state back to every root, applies the reset, restarts the integrator from the modified state and
    continues to the last evaluation time — bit for bit the oracle integrating the generated host twin (the reference's reset test problem as DiffSL text;
    single IVP and a lock-step batch of identical members, which must agree on every event), and the same again for a run-time-sized hybrid model."""
    base = "in = [k]\nk { 0.1 }\nu_i { x = 1, y = 1 }\nF_i { -k * x, -k * y }\nstop_i { x - 0.6, x - 0.3 }\nreset_i { 0.4, 0.4 }\n"
    big = ("in = [k]\nk { 0.1 }\nu_i { (0:10): x = 1 }\nF_i { -k * x_i }\nstop_i { x_i[0:1] - 0.6, x_i[0:1] - 0.3 }\nreset_i { (0:10): 0.4 }\n")
    t_eval = [0.0, 2.0, 5.2, 7.9, 8.0, 12.0, 16.0]
    hm, om = METHOD[method], {"bdf": O.METHOD_BDF, "tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
    for code, n in ((base, 2), (big, 10)):
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        for nb in (1, 3):
            p = np.full((nb, 1), 0.1)
            # the host-driven path is what this test is about: asked for explicitly, because AUTO now takes the run-time-sized banded model's lane-per-member BDF, which
            # carries the reset through the events inside the launch (round 4, tests/test_gpu_resident_reset.py)
            s = H.Solver(m, p, nbatch=nb, method=hm, rtol=1e-6, atol=[1e-6], ensemble_mode=0)
            assert s.ensemble_mode()[1] == 0
            y, reason = s.solve_dense(t_eval)  # [nt, nbatch, n]
            yo, so, failed = O.solve_dense_independent(mid, p, t_eval, group=nb, method=om, rtol=1e-6, atol=[1e-6])
            assert failed == 0 and reason == 2  # TstopReached: every evaluation time is filled
            assert np.array_equal(np.transpose(y, (1, 0, 2)), yo)
            t0, per = -np.log(0.6) / 0.1, np.log(4.0 / 3.0) / 0.1  # closed-form sawtooth: decay to 0.6, then 0.4 -> 0.3 periods
            assert abs(yo[0, -1, 0] - 0.4 * np.exp(-0.1 * ((t_eval[-1] - t0) % per))) < 1e-4 and yo.shape[2] == n
            assert s.stats()["number_of_steps"] == so[0, 0]
    # hybrid DAE: apply_reset_with_mass (state.rs:279-306) makes the reset state consistent with a Newton solve on InitOp, without line search
    dae = ("in = [k]\nk { 0.1 }\nu_i { x = 1, z = 2 }\ndudt_i { dxdt = 0, dzdt = 0 }\nM_i { dxdt, 0 }\nF_i { -k * x, z * z - 4 * x * x }\n"
           "stop_i { x - 0.6, x - 0.3 }\nreset_i { 0.4, 0.81 }\n")
    m, mid = fe.DiffslModel(dae), D.host_model(O, dae)
    for nb in (1, 3):
        p = np.full((nb, 1), 0.1)
        s = H.Solver(m, p, nbatch=nb, method=hm, rtol=1e-6, atol=[1e-6])
        y, reason = s.solve_dense(t_eval)
        yo, so, failed = O.solve_dense_independent(mid, p, t_eval, group=nb, method=om, rtol=1e-6, atol=[1e-6])
        assert failed == 0 and reason == 2 and np.array_equal(np.transpose(y, (1, 0, 2)), yo)
        assert abs(yo[0, -1, 1] - 2.0 * yo[0, -1, 0]) < 1e-5 and s.stats()["number_of_steps"] == so[0, 0]


def test_the_model_index_of_a_diffsl_text_reaches_the_device_model(H, O, fe):
    """`N` (the reference's DiffSlContext::model_index) is a compile-time constant of the generated model: the device model compiled for index 2 integrates like the
    host twin compiled for index 2, bit for bit, and unlike the one for index 0."""
    code = "in = [k]\nk { 0.5 }\nu_i { x = 1, y = 2 }\nF_i { -(N + 1) * k * x, -k * y + N }\n"
    p = 0.25 + 0.05 * np.arange(70)[:, None]
    te = [0.5, 1.0, 2.0]
    tol = dict(rtol=1e-6, atol=[1e-8])
    ys = {}
    for idx in (0, 2):
        m, mid = fe.DiffslModel(code, model_index=idx), D.host_model(O, code, model_index=idx)
        O.set_det_pow(True)
        try:
            y, tot = H.Solver(m, p, nbatch=len(p), **tol).solve_dense_adaptive(te, group=1)
            yo, so, failed = O.solve_dense_independent(mid, p, te, nthreads=4, **tol)
        finally:
            O.set_det_pow(False)
        assert failed == 0 and np.array_equal(y, np.transpose(yo, (1, 0, 2)))
        assert np.allclose(y[:, :, 0], np.exp(-(idx + 1) * p[None, :, 0] * np.asarray(te)[:, None]), rtol=1e-4, atol=1e-7)
        ys[idx] = y
    assert not np.array_equal(ys[0], ys[2])


def test_large_sparse_models_assemble_their_dense_jacobian_from_the_structural_nonzeros(H, O, fe, monkeypatch):
    """Round 4: the dense Jacobian of a run-time-sized model with n >= 128 whose f_y is sparse (the reference's 962-state DFN model: 9 Jacobians x 14 ms of a 0.77 s
    solve) is assembled from its structural nonzeros alone (k_jit_dyn_jacobian_sparse: the front end lists them, the matrix is zeroed first) — entry by entry the
    evaluation of the dense form, so the two matrices are equal (up to the sign of structural zeros) and equal to the oracle's J e_j."""
    from diffsol_amd import _ffi
    n = 150
    # a reaction-diffusion chain with a nonlinear source and one long-range coupling per row: bandwidth ~n/2, 5 nonzeros per row
    rows = ",\n".join([f"  ({i},{i - 1}): 1.0" for i in range(1, n)] + [f"  ({i},{i}): -2.0" for i in range(n)] + [f"  ({i},{i + 1}): 1.0" for i in range(n - 1)])
    far = ",\n".join(f"  ({i},{(i + n // 2) % n}): 0.25" for i in range(n))
    code = (f"in = [d, k]\nd {{ 1.0 }}\nk {{ 0.5 }}\nA_ij {{\n{rows}\n}}\nB_ij {{\n{far}\n}}\nu_i {{ (0:{n}): 0.3 }}\nlap_i {{ A_ij * u_j }}\nfar_i {{ B_ij * u_j }}\n"
            f"F_i {{ d * lap_i + k * u_i * (1 - u_i) + far_i * u_i }}\n")
    src = fe.generate(code, fe.TARGET_HIP_DYNAMIC)[0]
    assert "DSH_JIT_JAC_NNZ" in src
    m, mid = fe.DiffslModel(code), D.host_model(O, code)
    L = _ffi.load_device_lib()
    nb, t = 9, 0.2
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(n)
    x, p = rng.uniform(0.1, 0.9, (nb, n)), rng.uniform(0.5, 2.0, (nb, 2))
    X, P = H.HipVec.from_vec(x, c), H.HipVec.from_vec(p, c)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSH_JAC_SPARSE", mode)
        J = H.HipMat.zeros(n, n, c)
        assert L.dsh_model_jacobian(c._h, m.model_id, 0, nb, t, X.ptr, P.ptr, J.ptr) == 0
        out[mode] = J.to_array()
    assert np.array_equal(out["1"], out["0"]) and (out["1"] != 0).sum() <= nb * 5 * n
    jref = np.empty((nb, n, n))
    for b in range(nb):
        for j in range(n):
            e = np.zeros(n); e[j] = 1.0
            jref[b, :, j] = O.model_jac_mul(mid, x[b], p[b], e, t)
    assert np.array_equal(out["1"], jref)
    # and a solve through it (host-driven lock-step over the dense LU) equals the oracle's
    s = H.Solver(m, p, nbatch=nb, rtol=1e-6, atol=[1e-8], ensemble_mode=0)
    y, _ = s.solve_dense([0.05, 0.2])
    yo, so, failed = O.solve_dense_independent(mid, p, [0.05, 0.2], group=nb, rtol=1e-6, atol=[1e-8])
    assert failed == 0 and np.array_equal(np.transpose(y, (1, 0, 2)), yo)


@pytest.mark.parametrize("m_pairs,method", [(3, "bdf"), (4, "tr_bdf2"), (3, "esdirk34")])
def test_static_model_with_5_to_8_states_runs_per_member_through_its_run_time_sized_twin(H, O, fe, det_pow, m_pairs, method):
    """VERDICT r4 missing 6: a DiffSL model with 5 <= n <= 8 states is compiled in the static form by default (fused host-driven kernels), whose device-resident
    integrators stop at n = 4.  The front end registers the same model in the run-time-sized form (dsh_model_set_member_twin_source) and the first per-member request
    compiles it (dsh_model_member_twin): solve_dense_adaptive(group = 1) then runs on the wavefront-per-member kernels, bit-identical to independent CPU solves."""
    code = D.oscillators(m_pairs)  # n = 2 m_pairs = 6 or 8, dense coupling
    m = fe.DiffslModel(code)
    assert m.form == fe.FORM_STATIC and 5 <= m.n <= 8
    L = m._L
    assert L.dsh_model_has_wave_member(m.model_id, 0) == 0  # the static form itself has no such kernel ...
    nb = 70
    rng = np.random.default_rng(11 + m_pairs)
    p = np.stack([rng.uniform(30.0, 70.0, nb), rng.uniform(0.5, 2.0, nb), rng.uniform(0.01, 0.1, nb)], axis=1)
    tol = dict(rtol=1e-6, atol=[1e-8])
    s = H.Solver(m, p, nbatch=nb, method=METHOD[method], **tol)
    t_eval = [0.05, 0.1, 0.3]
    y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    tw = L.dsh_model_member_twin(m.model_id)
    assert tw >= 1000 and tw != m.model_id and L.dsh_model_has_wave_member(tw, 0) == 1  # ... its twin has, and it exists now
    mid = D.host_model(O, code)
    yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=1, method={"bdf": 0, "tr_bdf2": 1, "esdirk34": 2}[method], **tol)
    assert failed == 0 and tot["failed_members"] == 0 and (mem["status"] == 0).all()
    assert np.array_equal(mem["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)))
    # the automatic mode keeps the model on the kernels it was compiled for
    y2, _ = s.solve_dense(t_eval)
    assert np.isfinite(y2).all()
