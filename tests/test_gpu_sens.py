"""Forward sensitivities on the HIP backend: problem.bdf_sens() — Bdf::sensitivity_solve (bdf.rs:934-989), SensEquations
(ode_equations/sens_equations.rs), the sensitivity difference arrays in _update_step_size / _update_diff, sensitivities in the error control — through the
C ABI (dshs_create_sens, dshs_interpolate_sens; device operators dsh_model_rhs_sens / dsh_model_init_sens).  Everything bit for bit against the CPU oracle,
which reproduces all 13 counters of the reference's bdf_test_nalgebra_exponential_decay_sens snapshot (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from helpers import ORACLE_MODEL, robertson_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


def _points(s, pts):
    ys, ss = [], []
    for t in pts:
        while abs(s.scalars()[0] if hasattr(s, "scalars") else s.state()["t"]) < abs(t):
            s.step()
        ys.append(s.interpolate(t))
        ss.append(s.interpolate_sens(t))
    return np.array(ys), np.array(ss)


def test_device_sensitivity_operators_match_the_oracle_models(H, O):
    """dsh_model_rhs_sens / dsh_model_init_sens: df/dp and dy0/dp as n x np matrices in one launch = the oracle's sens_mul / init_sens_mul column by column."""
    from diffsol_amd import _ffi
    import ctypes as C
    L = _ffi.load_device_lib()
    assert L.dsh_model_has_sens(H.MODELS["robertson_ode"], 1) == 1 and L.dsh_model_has_sens(H.MODELS["exponential_decay"], 0) == 1
    assert L.dsh_model_has_sens(H.MODELS["robertson"], 0) == 1 and L.dsh_model_has_sens(H.MODELS["exponential_decay_with_algebraic"], 0) == 1
    assert L.dsh_model_has_sens(H.MODELS["rlc"], 0) == 0 and L.dsh_model_has_sens(H.MODELS["heat1d"], 16) == 0
    nb = 37
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(2)
    for name, size, n, npar in (("robertson_ode", 1, 3, 3), ("exponential_decay", 0, 2, 2)):
        x, p = rng.uniform(0.1, 2.0, (nb, n)), rng.uniform(0.5, 3.0, (nb, npar))
        X, P = H.HipVec.from_vec(x, c), H.HipVec.from_vec(p, c)
        S = H.HipMat.zeros(n, npar, c)
        assert L.dsh_model_rhs_sens(c._h, H.MODELS[name], size, nb, 0.3, X.ptr, P.ptr, S.ptr) == 0
        got = S.to_array()  # [nb, n, np]
        if name == "robertson_ode":
            ref = np.stack([np.stack([-x[:, 0], x[:, 0], 0 * x[:, 0]], 1), np.stack([x[:, 1] * x[:, 2], -(x[:, 1] * x[:, 2]), 0 * x[:, 0]], 1),
                            np.stack([0 * x[:, 0], -(x[:, 1] * x[:, 1]), x[:, 1] * x[:, 1]], 1)], 2)
        else:
            ref = np.stack([np.stack([-x[:, 0], -x[:, 1]], 1), np.zeros((nb, 2))], 2)
        assert np.array_equal(got, ref), name
        assert L.dsh_model_init_sens(c._h, H.MODELS[name], size, nb, 0.0, P.ptr, S.ptr) == 0
        ref0 = np.zeros((nb, n, npar))
        if name == "exponential_decay":
            ref0[:, :, 1] = 1.0
        assert np.array_equal(S.to_array(), ref0)
    S = H.HipMat.zeros(4, 6, c)
    assert L.dsh_model_rhs_sens(c._h, H.MODELS["rlc"], 0, nb, 0.0, X.ptr, P.ptr, S.ptr) < 0  # loud: this model has no parameter derivatives


@pytest.mark.parametrize("error_control", [True, False])
def test_exponential_decay_sensitivities_equal_the_oracle_bitwise_and_reproduce_the_reference_counters(H, O, error_control):
    """The reference's own sensitivity test problem (exponential_decay_problem_sens): single IVP first — the HIP path gives the 13 counters of the reference's
    insta snapshot (bdf.rs:1815-1834) — then a batched ensemble in lock-step, states / sensitivities / counters bit for bit against the oracle."""
    pts = [float(i) for i in range(10)]
    skw = dict(sens=True, sens_rtol=1e-6, sens_atol=[1e-6, 1e-6]) if error_control else dict(sens=True)
    s = H.Solver("exponential_decay", [[0.1, 1.0]], nbatch=1, rtol=1e-6, atol=[1e-6], **skw)
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay"], [0.1, 1.0], rtol=1e-6, atol=[1e-6], **skw)
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and s.stats() == o.stats()
    if error_control:
        st = s.stats()
        assert [st[k] for k in st] == [14, 56, 1, 175, 0, 1, 0, 0, 1, 12, 60, 123, 2]
    t = np.array(pts)[:, None]
    tol = 2e-5 if error_control else 5e-3  # outside the error control nothing bounds the effect of the reference's c = 0 start on the sensitivities
    assert np.abs(ss[:, 1, 0, :] - np.exp(-0.1 * t)).max() < tol and np.abs(ss[:, 0, 0, :] + t * np.exp(-0.1 * t)).max() < tol
    nb = 5
    p = np.stack([0.1 * (np.arange(nb) + 1), 1.0 + np.arange(nb)], axis=1)
    sb = H.Solver("exponential_decay", p, nbatch=nb, rtol=1e-6, atol=[1e-6], **skw)
    ob = O.OracleSolver(ORACLE_MODEL["exponential_decay"], p, nbatch=nb, rtol=1e-6, atol=[1e-6], **skw)
    yb, sbv = _points(sb, pts[:6])
    yob, sob = _points(ob, pts[:6])
    assert np.array_equal(yb, yob) and np.array_equal(sbv, sob) and sb.stats() == ob.stats()
    assert np.array_equal(sb.interpolate_sens(), ob.interpolate_sens())  # state.s


def test_robertson_ensemble_sensitivities_equal_the_oracle_bitwise_and_finite_differences(H, O):
    """Config-2-shaped ensemble (Robertson ODE sweep) with forward sensitivities with respect to (k1, k2, k3), in the error control: lock-step ensemble on the GPU
    = oracle bit for bit (states, 3 x n sensitivities per member, all counters); the sensitivities agree with central differences of plain solves."""
    nb = 6
    p = robertson_params(nb, seed=77)
    kw = dict(model_size=1, rtol=1e-5, atol=[1e-9, 1e-12, 1e-9])
    skw = dict(sens=True, sens_rtol=1e-5, sens_atol=[1e-8])
    pts = [0.4, 4.0, 40.0]
    s = H.Solver("robertson_ode", p, nbatch=nb, **kw, **skw)
    assert not s.fused  # the sensitivity path runs on the trait operations
    o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, nbatch=nb, **kw, **skw)
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and s.stats() == o.stats()
    tight = dict(model_size=1, rtol=1e-10, atol=[1e-14, 1e-16, 1e-14])
    st = H.Solver("robertson_ode", p, nbatch=nb, sens=True, sens_rtol=1e-10, sens_atol=[1e-12], **tight)
    _, s_t = _points(st, pts)
    for j in range(3):
        dp = 1e-5 * p[:, j]
        pp, pm = p.copy(), p.copy()
        pp[:, j] += dp
        pm[:, j] -= dp
        yp, _ = H.Solver("robertson_ode", pp, nbatch=nb, **tight).solve_to_points(pts)
        ym, _ = H.Solver("robertson_ode", pm, nbatch=nb, **tight).solve_to_points(pts)
        fd = (yp - ym) / (2 * dp)[None, :, None]
        assert np.abs(s_t[:, j] - fd).max() <= 5e-4 * np.abs(fd).max(), j


def test_dae_sensitivities_reproduce_the_reference_snapshots_and_equal_the_oracle_bitwise(H, O):
    """Singular mass matrix: set_consistent_augmented's InitOp over the sensitivity equations (state.rs:187-238) and the mass matrix in the sensitivity
    residual, on the device.  The HIP path gives all 13 counters of test_bdf_nalgebra_exponential_decay_algebraic_sens (bdf.rs:2118-2141) and of
    test_bdf_nalgebra_robertson_sens (bdf.rs:2248-2271: 319 steps, 28 failed nonlinear solves) and the oracle's states and sensitivities bit for bit;
    then a Robertson DAE ensemble in lock-step against the oracle."""
    kw = dict(rtol=1e-6, atol=[1e-6], sens=True, sens_rtol=1e-6, sens_atol=[1e-6, 1e-6, 1e-6])
    s = H.Solver("exponential_decay_with_algebraic", [[0.1]], nbatch=1, **kw)
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay_with_algebraic"], [0.1], **kw)
    pts = [i / 10.0 for i in range(10)]
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    st = s.stats()
    assert [st[k] for k in st] == [24, 45, 8, 115, 0, 1, 0, 0, 8, 15, 66, 64, 3]
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and st == o.stats()
    t = np.array(pts)[:, None]
    assert np.abs(ss[:, 0, 0, :] + t * np.exp(-0.1 * t)).max() < 2e-5
    kw = dict(rtol=1e-4, atol=[1e-8, 1e-6, 1e-6], sens=True)
    pts = [0.4, 4.0, 40.0, 400.0, 4000.0, 4e4, 4e5, 4e6, 4e7, 4e8, 4e9, 4e10]
    s = H.Solver("robertson", [[0.04, 1.0e4, 3.0e7]], nbatch=1, options=dict(max_nonlinear_solver_failures=70), **kw)
    o = O.OracleSolver(ORACLE_MODEL["robertson"], [0.04, 1.0e4, 3.0e7], options=dict(max_nonlinear_solver_failures=70), **kw)
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    st = s.stats()
    assert [st[k] for k in st] == [92, 319, 4, 1941, 28, 1, 26, 2, 4, 59, 575, 1522, 31]
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and st == o.stats()
    nb = 5
    p = robertson_params(nb, seed=3)
    sb = H.Solver("robertson", p, nbatch=nb, sens_rtol=1e-4, sens_atol=[1e-8, 1e-6, 1e-6], **kw)
    ob = O.OracleSolver(ORACLE_MODEL["robertson"], p, nbatch=nb, sens_rtol=1e-4, sens_atol=[1e-8, 1e-6, 1e-6], **kw)
    yb, sv = _points(sb, pts[:4])
    yob, sov = _points(ob, pts[:4])
    assert np.array_equal(yb, yob) and np.array_equal(sv, sov) and sb.stats() == ob.stats()
    assert np.abs(sv.sum(axis=-1)).max() < 1e-6 * np.abs(sv).max()  # x + y + z = 1: every sensitivity sums to zero


def test_sensitivity_requests_that_the_backend_cannot_serve_fail_loudly(H):
    with pytest.raises(H.DiffsolHipError):
        H.Solver("rlc", [[100.0, 1.0, 1e-3, 10.0, 100.0, 1e3]] * 2, nbatch=2, rtol=1e-4, atol=[1e-6], sens=True)        # a model without parameter derivatives
    s = H.Solver("robertson_ode", robertson_params(2), nbatch=2, model_size=1)
    with pytest.raises(H.DiffsolHipError):
        s.interpolate_sens(0.0)                                                                                         # not created with sensitivities


def test_diffsl_models_integrate_their_parameter_sensitivities_on_the_device(H, O):
    """DiffSL models with inputs carry sens_mul / init_sens_mul (forward-mode differentiation in parameter space by the front end, host/diffsl.hpp) in both
    device forms.  Register-resident form: the reference's own DiffSL sensitivity problem (text of exponential_decay_problem_diffsl) on the HIP path gives all
    13 counters of bdf_test_nalgebra_exponential_decay_diffsl_sens (bdf.rs:1845-1862) and the bits of the oracle integrating the generated host twin.
    Run-time-sized form (n = 12, one thread per component): a decay chain with parameter-dependent initial values, a batched lock-step ensemble, bit for bit
    against the oracle; sensitivities against central differences of plain solves."""
    import diffsl_models as D
    from diffsol_amd import diffsl as fe
    code = "in_i { k = 0.1, y0 = 1.0 }\nu_i { x = y0, y = y0 }\nF_i { -k * u_i }\nout_i { u_i }\n"
    m, mid = fe.DiffslModel(code), D.host_model(O, code)
    assert m.form == fe.FORM_STATIC
    kw = dict(rtol=1e-6, atol=[1e-6], sens=True, sens_rtol=1e-6, sens_atol=[1e-6, 1e-6])
    s = H.Solver(m, [[0.1, 1.0]], nbatch=1, **kw)
    o = O.OracleSolver(mid, [0.1, 1.0], **kw)
    pts = [float(i) for i in range(10)]
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    st = s.stats()
    assert [st[k] for k in st] == [14, 56, 1, 175, 0, 1, 0, 0, 1, 12, 60, 123, 2]
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and st == o.stats()
    chain = ("in = [k, c]\nk { 0.5 } c { 1.0 }\nu_i { (0): a = c * c, (1:12): b = 0.1 * c }\n"
             "A_ij { (0..12, 0..12): -1.0, (1..12, 0..11): 1.0 }\nF_i { k * A_ij * u_j }\n")
    m, mid = fe.DiffslModel(chain), D.host_model(O, chain)
    assert m.form == fe.FORM_DYNAMIC
    nb = 4
    p = np.stack([0.3 + 0.2 * np.arange(nb), 1.0 + 0.5 * np.arange(nb)], axis=1)
    kw = dict(rtol=1e-7, atol=[1e-9], sens=True, sens_rtol=1e-7, sens_atol=[1e-9])
    sb = H.Solver(m, p, nbatch=nb, **kw)
    ob = O.OracleSolver(mid, p, nbatch=nb, **kw)
    pts = [0.5, 2.0, 6.0]
    yb, sv = _points(sb, pts)
    yob, sov = _points(ob, pts)
    assert np.array_equal(yb, yob) and np.array_equal(sv, sov) and sb.stats() == ob.stats()
    tight = dict(rtol=1e-11, atol=[1e-13])
    for j in range(2):
        dp = 1e-6 * p[:, j]
        pp, pm = p.copy(), p.copy()
        pp[:, j] += dp
        pm[:, j] -= dp
        yp, _ = H.Solver(m, pp, nbatch=nb, **tight).solve_to_points(pts)
        ym, _ = H.Solver(m, pm, nbatch=nb, **tight).solve_to_points(pts)
        fd = (yp - ym) / (2 * dp)[None, :, None]
        assert np.abs(sv[:, j] - fd).max() <= 2e-5 * np.abs(fd).max(), j
    noin = fe.DiffslModel("u_i { x = 1 }\nF_i { -x }\n")
    with pytest.raises(H.DiffsolHipError):
        H.Solver(noin, [[0.0]], nbatch=1, sens=True)  # no inputs: nothing to differentiate with respect to


@pytest.mark.parametrize("method,model,snap", [
    ("tr_bdf2", "exp", [10, 90, 0, 620, 0, 1, 0, 0, 0, 9, 207, 421, 2]),
    ("esdirk34", "exp", [6, 33, 0, 347, 0, 1, 0, 0, 0, 5, 107, 246, 1]),
    ("tr_bdf2", "robertson", [77, 286, 0, 4146, 30, 1, 29, 1, 0, 46, 1303, 2954, 34]),
    ("esdirk34", "robertson", [68, 333, 0, 6856, 10, 1, 8, 2, 0, 57, 2272, 4644, 17])])
def test_sdirk_sensitivities_reproduce_the_reference_snapshots_and_equal_the_oracle_bitwise(H, O, method, model, snap):
    """problem.tr_bdf2_sens() / esdirk34_sens() on the HIP backend (host/sdirk.hpp: the sensitivity half of every stage, sensitivities in the error norm, the
    Checkpoint linearisation of new_augmented): all 13 counters of the reference's four SDIRK sensitivity snapshots (sdirk.rs:707-730, :782-805, :894-918,
    :945-968 — the last two on the Robertson DAE), states and sensitivities bit for bit the oracle's; then a batched lock-step ensemble against the oracle."""
    hm = H.METHOD_TR_BDF2 if method == "tr_bdf2" else H.METHOD_ESDIRK34
    om = O.METHOD_TR_BDF2 if method == "tr_bdf2" else O.METHOD_ESDIRK34
    if model == "exp":
        name, p1, kw, pts, okw = "exponential_decay", [0.1, 1.0], dict(rtol=1e-6, atol=[1e-6], sens=True, sens_rtol=1e-6, sens_atol=[1e-6, 1e-6]), [float(i) for i in range(10)], {}
    else:
        okw = dict(options=dict(max_nonlinear_solver_iterations=10)) if method == "tr_bdf2" else {}
        name, p1, kw, pts = "robertson", [0.04, 1.0e4, 3.0e7], dict(rtol=1e-4, atol=[1e-8, 1e-6, 1e-6], sens=True), [0.4, 4.0, 40.0, 400.0, 4000.0, 4e4, 4e5, 4e6, 4e7, 4e8, 4e9, 4e10]
    s = H.Solver(name, [p1], nbatch=1, method=hm, **kw, **okw)
    o = O.OracleSolver(ORACLE_MODEL[name], p1, method=om, **kw, **okw)
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    st = s.stats()
    assert [st[k] for k in st] == snap
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and st == o.stats()
    nb = 4
    p = np.stack([0.1 * (np.arange(nb) + 1), 1.0 + np.arange(nb)], axis=1) if model == "exp" else robertson_params(nb, seed=5)
    sb = H.Solver(name, p, nbatch=nb, method=hm, **kw, **okw)
    ob = O.OracleSolver(ORACLE_MODEL[name], p, nbatch=nb, method=om, **kw, **okw)
    yb, sv = _points(sb, pts[:4])
    yob, sov = _points(ob, pts[:4])
    assert np.array_equal(yb, yob) and np.array_equal(sv, sov) and sb.stats() == ob.stats()
