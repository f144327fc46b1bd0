"""Forward sensitivities INSIDE the device-resident BDF (dsh_bdf_solve_adaptive_sens, k_bdf_adaptive<.., SENS>): states and dy/dp_j of every
parameter at t_eval from one launch, against the oracle's solve_dense_sensitivities per member (group = 1) and per 64-member lock-step group (group = 64).

What is restated on the device: new_with_sensitivities_and_consistent (state.rs:1032-1083), Bdf::new_augmented (bdf.rs:370-432, the sensitivity operator's c = 0
until the first step-size change), sensitivity_solve (:934-989: one Newton solve per parameter with the factors of the state equations and the SHARED
Convergence), the sensitivity terms of error_control (:844-858) and predict_error_control (:871-932), _update_step_size's chain of every sensitivity difference
array through the one scratch matrix (:546-548), interpolate_sens (:1162-1215).  With the deterministic pow on both sides every counter and every output bit agrees."""
import numpy as np
import pytest

from helpers import ORACLE_MODEL
from bench import robertson_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


@pytest.fixture
def det_pow(O):
    O.set_det_pow(True)
    yield
    O.set_det_pow(False)


T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]
ROB = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])


def _pair(H, O, model, p, t_eval, size, group, sens_tol, expect_failed_at_most=0, **tol):
    nb = len(p)
    kw = dict(sens_rtol=sens_tol[0], sens_atol=sens_tol[1]) if sens_tol else {}
    s = H.Solver(model, p, nbatch=nb, model_size=size, sens=True, **kw, **tol)
    y, sens, tot, m = s.solve_dense_adaptive_sens(t_eval, group=group, want_member_stats=True)
    yo, so, sto, failed = O.solve_dense_independent_sens(ORACLE_MODEL[model], np.asarray(p, dtype=float), t_eval, model_size=size, nthreads=8, group=group, **kw, **tol)
    # a member (group) that fails — too many error-test failures under tight sensitivity tolerances — fails in both, at the same step: the counters, the columns
    # written up to there and the NaN columns behind are compared like everything else
    assert failed == tot["failed_members"] == int((m["status"] != 0).sum()) and failed <= expect_failed_at_most
    assert np.array_equal(m["stats"].T, sto), "counters differ"
    assert np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True), "states differ"
    assert np.array_equal(sens, np.transpose(so, (0, 2, 1, 3)), equal_nan=True), "sensitivities differ"
    assert tot["number_of_steps"] == int(sto[:, 0].sum()) and tot["number_of_nonlinear_solver_iterations"] == int(sto[:, 1].sum())
    _pair.failed = failed
    return y, sens, sto


@pytest.mark.parametrize("group", [1, 64])
@pytest.mark.parametrize("error_control", [None, "mild", "tight"])
def test_resident_bdf_with_sensitivities_is_bit_identical_to_the_oracle_on_robertson(H, O, det_pow, group, error_control):
    """Robertson ODE with its 3 rate constants as parameters (test_models/robertson_ode_with_sens.rs): 300 members (4 full wavefronts + a partial one).
    "tight": the states' absolute tolerances on sensitivities that are 1e4 times larger — some members exhaust their error-test failures, in the oracle too."""
    p = robertson_params(300, seed=11)
    tol = {None: None, "mild": (1e-4, [1e-6]), "tight": (1e-4, [1e-8, 1e-14, 1e-6])}[error_control]
    y, sens, st = _pair(H, O, "robertson_ode", p, T_EVAL, 1, group, tol, expect_failed_at_most=0 if error_control != "tight" else 128, **ROB)
    # (that the oracle's sensitivities are the derivatives of its states: tests/test_oracle_sens_independent.py)
    if error_control != "tight":
        assert np.isfinite(sens).all() and np.abs(sens[0]).max() > 1.0
    if error_control == "mild":  # the sensitivities take part in the error test: more work than without
        _, _, st0 = _pair(H, O, "robertson_ode", p[:64], T_EVAL, 1, group, None, **ROB)
        assert st[:64, 0].sum() > st0[:, 0].sum()


@pytest.mark.parametrize("group", [1, 64])
def test_resident_bdf_with_sensitivities_on_the_exponential_decay_snapshot_problem(H, O, det_pow, group):
    """exponential_decay_problem_sens (test_models/exponential_decay.rs:224-262): y' = -k y, y(0) = y0, parameters (k, y0); rtol = atol = 1e-6, sensitivities in
    the error control.  Analytic: y = y0 e^{-kt}, dy/dk = -t y0 e^{-kt}, dy/dy0 = e^{-kt}."""
    k = 0.1 * (1 + np.arange(130) % 7)
    y0 = 1.0 + 0.25 * (np.arange(130) % 5)
    p = np.stack([k, y0], axis=1)
    te = [float(i) for i in range(0, 11)]
    y, sens, st = _pair(H, O, "exponential_decay", p, te, 0, group, (1e-6, [1e-6]), rtol=1e-6, atol=[1e-6, 1e-6])
    t = np.asarray(te)[:, None]
    e = np.exp(-k[None, :] * t)
    assert np.allclose(y[:, :, 0], y0[None, :] * e, rtol=2e-4, atol=1e-6) and np.allclose(y[:, :, 1], y[:, :, 0])
    assert np.allclose(sens[0, :, :, 0], -t * y0[None, :] * e, rtol=1e-3, atol=3e-5)
    assert np.allclose(sens[1, :, :, 0], e, rtol=1e-3, atol=3e-5)
    # the reference's snapshot member (k = 0.1, y0 = 1): the device-resident run of that one problem repeats the counters of the oracle's stepping solver
    _pair(H, O, "exponential_decay", [[0.1, 1.0]], te, 0, 1, (1e-6, [1e-6]), rtol=1e-6, atol=[1e-6, 1e-6])


@pytest.mark.parametrize("group", [1, 64])
@pytest.mark.parametrize("method", ["tr_bdf2", "esdirk34"])
@pytest.mark.parametrize("error_control", [False, True])
def test_resident_sdirk_with_sensitivities_is_bit_identical_to_the_oracle(H, O, det_pow, method, group, error_control):
    """TR-BDF2 and ESDIRK34 (k_sdirk_resident<.., SENS>, dsh_sdirk_solve_resident_sens): the sensitivity half of do_stage_sdirk (runge_kutta.rs:691-748: per stage and
    parameter a Newton solve with the factors of the state equations and the shared Convergence, SensRhs linearised about the stage's state), the linearisation made
    at construction (sdirk.rs:251), the sensitivity terms of the error norm (:812-822), interpolate_sens (:1237-1330: TR-BDF2's Hermite interpolant, ESDIRK34's beta
    polynomial), against the oracle's solve_dense_sensitivities per member / per 64-member group."""
    hm = {"tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
    om = {"tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
    kw = dict(sens_rtol=1e-4, sens_atol=[1e-6]) if error_control else {}
    for model, size, p, te, tol in (("robertson_ode", 1, robertson_params(200, seed=7), T_EVAL[:6], ROB),
                                    ("exponential_decay", 0, np.stack([0.1 * (1 + np.arange(70) % 7), 1.0 + 0.25 * (np.arange(70) % 5)], axis=1), [float(i) for i in range(0, 10)],
                                     dict(rtol=1e-6, atol=[1e-6, 1e-6]))):
        nb = len(p)
        s = H.Solver(model, p, nbatch=nb, model_size=size, method=hm, sens=True, **kw, **tol)
        y, sens, tot, m = s.solve_dense_adaptive_sens(te, group=group, want_member_stats=True)
        yo, so, sto, failed = O.solve_dense_independent_sens(ORACLE_MODEL[model], np.asarray(p, dtype=float), te, model_size=size, nthreads=8, group=group, method=om, **kw, **tol)
        assert failed == tot["failed_members"] == int((m["status"] != 0).sum()) == 0
        assert np.array_equal(m["stats"].T, sto), "counters differ"
        assert np.array_equal(y, np.transpose(yo, (1, 0, 2))), "states differ"
        assert np.array_equal(sens, np.transpose(so, (0, 2, 1, 3))), "sensitivities differ"
    k, y0 = p[:, 0], p[:, 1]  # exponential decay: analytic derivatives
    t = np.asarray(te)[:, None]
    e = np.exp(-k[None, :] * t)
    assert np.allclose(sens[0, :, :, 0], -t * y0[None, :] * e, rtol=2e-3, atol=2e-4) and np.allclose(sens[1, :, :, 0], e, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("group", [1, 64])
@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
def test_resident_integrators_with_sensitivities_of_a_diffsl_model(H, O, det_pow, group, method):
    """A DiffSL model with inputs in its register-resident form carries sens_mul / init_sens_mul (forward-mode differentiation by the front end): the same kernel
    template, instantiated by hiprtc for the user's model with SENS = true.  Robertson's kinetics written in DiffSL with the three rate constants as inputs, and
    the reference's exponential_decay_problem_diffsl text (parameter-dependent initial state); the checker is the oracle integrating the generated host twin."""
    import diffsl_models as D
    from diffsol_amd import diffsl as fe
    for code, p, te, tol, stol in (
            (D.ROBERTSON_ODE, robertson_params(150, seed=4), T_EVAL[:6], ROB, (1e-4, [1e-6])),
            ("in_i { k = 0.1, y0 = 1.0 }\nu_i { x = y0, y = y0 }\nF_i { -k * u_i }\nout_i { u_i }\n",
             np.stack([0.1 * (1 + np.arange(70) % 7), 1.0 + 0.25 * (np.arange(70) % 5)], axis=1), [float(i) for i in range(0, 10)], dict(rtol=1e-6, atol=[1e-6, 1e-6]), (1e-6, [1e-6]))):
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        assert m.form == fe.FORM_STATIC
        nb = len(p)
        hm = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
        om = {"bdf": O.METHOD_BDF, "tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
        s = H.Solver(m, p, nbatch=nb, method=hm, sens=True, sens_rtol=stol[0], sens_atol=stol[1], **tol)
        y, sens, tot, mm = s.solve_dense_adaptive_sens(te, group=group, want_member_stats=True)
        yo, so, sto, failed = O.solve_dense_independent_sens(mid, np.asarray(p, dtype=float), te, nthreads=8, group=group, method=om, sens_rtol=stol[0], sens_atol=stol[1], **tol)
        assert failed == 0 and tot["failed_members"] == 0 and (mm["status"] == 0).all()
        assert np.array_equal(mm["stats"].T, sto) and np.array_equal(y, np.transpose(yo, (1, 0, 2))) and np.array_equal(sens, np.transpose(so, (0, 2, 1, 3)))


def test_resident_sensitivities_refuse_what_they_do_not_cover(H):
    import diffsol_amd
    dev = diffsol_amd._ffi.load_device_lib()
    assert dev.dsh_model_has_adaptive_sens(diffsol_amd.MODELS["robertson_ode"], 1) == 1
    assert dev.dsh_model_has_adaptive_sens(diffsol_amd.MODELS["exponential_decay"], 0) == 1
    assert dev.dsh_model_has_adaptive_sens(diffsol_amd.MODELS["robertson"], 0) == 0  # DAE (mass matrix): host-driven
    assert dev.dsh_model_has_adaptive_sens(diffsol_amd.MODELS["heat1d"], 20) == 0
    s = H.Solver("robertson", [[0.04, 1e4, 3e7]] * 4, nbatch=4, sens=True, rtol=1e-4, atol=[1e-8, 1e-6, 1e-6])
    with pytest.raises(Exception, match="forward sensitivities"):
        s.solve_dense_adaptive_sens([1.0])
    s2 = H.Solver("robertson_ode", [[0.04, 1e4, 3e7]] * 4, nbatch=4, model_size=1, **ROB)  # no sens requested
    with pytest.raises(Exception, match="dshs_create_sens"):
        s2.solve_dense_adaptive_sens([1.0])


@pytest.mark.parametrize("group", [1, 64])
@pytest.mark.parametrize("error_control", [None, (1e-6, [1e-6])])
def test_banded_lane_per_member_bdf_with_sensitivities_heat_and_battery(H, O, det_pow, group, error_control):
    """forward sensitivities in the banded lane-per-member form (bdf.rs:934-989 for the PDE / battery models).  The state, the difference arrays
    of the state AND of every sensitivity live in per-lane memory (k_bdf_adaptive's banded branch with SENS; the sensitivity solves run on the banded factors):
    heat1d with du/dD (n = 20 and 33: bandwidth 1) and the single-particle battery model with d(state)/dI (n = 42), 70 members (one full wavefront + a partial
    one), per member and in lock-step groups, with and without sensitivity error control — every counter and every bit of states and sensitivities against the
    oracle's solve_dense_sensitivities on the host twin."""
    import diffsl_models as D
    from diffsol_amd import diffsl as fe
    rng = np.random.default_rng(5)
    cases = [(D.heat1d(20), rng.uniform(0.5, 2.0, (70, 1)), [0.01, 0.05, 0.2], dict(rtol=1e-6, atol=[1e-7])),
             (D.heat1d(33), rng.uniform(0.5, 2.0, (70, 1)), [0.01, 0.1], dict(rtol=1e-5, atol=[1e-7])),
             (D.spm(20, no_stops=True), rng.uniform(0.6, 1.4, (70, 1)), [360.0, 1200.0, 3000.0], dict(rtol=1e-6, atol=[1e-6]))]
    for code, p, te, tol in cases:
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        assert m.form == fe.FORM_DYNAMIC and m.nroots == 0
        kw = dict(sens_rtol=error_control[0], sens_atol=error_control[1]) if error_control else {}
        s = H.Solver(m, p, nbatch=len(p), sens=True, **kw, **tol)
        y, sens, tot, mm = s.solve_dense_adaptive_sens(te, group=group, want_member_stats=True)
        yo, so, sto, failed = O.solve_dense_independent_sens(mid, np.asarray(p, dtype=float), te, nthreads=8, group=group, **kw, **tol)
        assert failed == 0 and tot["failed_members"] == 0 and (mm["status"] == 0).all()
        assert np.array_equal(mm["stats"].T, sto), "counters differ"
        assert np.array_equal(y, np.transpose(yo, (1, 0, 2))), "states differ"
        assert np.array_equal(sens, np.transpose(so, (0, 2, 1, 3))), "sensitivities differ"
        assert np.abs(sens).max() > 0


@pytest.mark.parametrize("method", ["tr_bdf2", "esdirk34"])
@pytest.mark.parametrize("group", [1, 64])
def test_banded_lane_per_member_sdirk_with_sensitivities(H, O, det_pow, method, group):
    """VERDICT r3 item 5, second half ("BDF first, then TR-BDF2"): the sensitivity half of do_stage_sdirk (runge_kutta.rs:691-748) in the banded lane-per-member form of
    k_sdirk_resident — the sensitivity stage solves run on the banded factors of the state equations — heat1d with du/dD and the battery model with d(state)/dI, per member
    and per 64-member group, with sensitivity error control: counters, states and sensitivities bit for bit against the oracle's solve_dense_sensitivities."""
    import diffsl_models as D
    from diffsol_amd import diffsl as fe
    hm = {"tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
    om = {"tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
    rng = np.random.default_rng(11)
    cases = [(D.heat1d(20), rng.uniform(0.5, 2.0, (70, 1)), [0.01, 0.05, 0.2], dict(rtol=1e-6, atol=[1e-7])),
             (D.spm(20, no_stops=True), rng.uniform(0.6, 1.4, (70, 1)), [360.0, 1200.0], dict(rtol=1e-6, atol=[1e-6]))]
    for code, p, te, tol in cases:
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        kw = dict(sens_rtol=1e-6, sens_atol=[1e-6])
        s = H.Solver(m, p, nbatch=len(p), sens=True, method=hm, **kw, **tol)
        y, sens, tot, mm = s.solve_dense_adaptive_sens(te, group=group, want_member_stats=True)
        yo, so, sto, failed = O.solve_dense_independent_sens(mid, np.asarray(p, dtype=float), te, nthreads=8, group=group, method=om, **kw, **tol)
        assert failed == 0 and tot["failed_members"] == 0 and (mm["status"] == 0).all()
        assert np.array_equal(mm["stats"].T, sto), "counters differ"
        assert np.array_equal(y, np.transpose(yo, (1, 0, 2))), "states differ"
        assert np.array_equal(sens, np.transpose(so, (0, 2, 1, 3))), "sensitivities differ"
        assert np.abs(sens).max() > 0


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
@pytest.mark.parametrize("error_control", [None, (1e-6, [1e-7])])
def test_wavefront_per_member_bdf_with_sensitivities_of_dense_models(H, O, det_pow, error_control, method):
    """VERDICT r3 missing 2, second half: forward sensitivities in the wavefront-per-member form (dense run-time-compiled models the register-resident and the banded
    lane forms do not cover).  k_bdf_wave_member<.., SENS> / k_sdirk_wave_member<.., SENS>: a component per lane, J s through the published vectors, the sensitivity solves on the state equations'
    factors, sdiff_j one row per lane.  Coupled oscillators with a dense coupling block (n = 12 and 20, three parameters) and a dense linear system with one
    parameter: every counter and every bit of states and sensitivities equal the oracle's solve_dense_sensitivities per member on the host twin."""
    import diffsl_models as D
    from diffsol_amd import diffsl as fe
    from diffsol_amd import _ffi
    rng = np.random.default_rng(8)
    nb = 70
    dense = ("in = [k]\nk { 1.0 }\nA_ij { (0:10, 0:10): -0.05, (0..10, 0..10): -1.0 }\nu_i { (0:10): 1.0 }\nlin_i { A_ij * u_j }\nF_i { k * lin_i + 0.1 * k * k }\n")
    cases = [(D.oscillators(6), np.stack([rng.uniform(20, 60, nb), rng.uniform(0.5, 2.0, nb), rng.uniform(0.005, 0.02, nb)], axis=1), [0.02, 0.1, 0.3], dict(rtol=1e-6, atol=[1e-8])),
             (D.oscillators(10), np.stack([rng.uniform(20, 60, nb), rng.uniform(0.5, 2.0, nb), rng.uniform(0.005, 0.02, nb)], axis=1), [0.05, 0.2], dict(rtol=1e-5, atol=[1e-7])),
             (dense, rng.uniform(0.5, 2.0, (nb, 1)), [0.1, 0.7, 2.0], dict(rtol=1e-7, atol=[1e-9]))]
    L = _ffi.load_device_lib()
    for code, p, te, tol in cases:
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        assert m.form == fe.FORM_DYNAMIC and m.lane_model_id is None and L.dsh_model_has_wave_member_sens(m.model_id, 0) == 1
        kw = dict(sens_rtol=error_control[0], sens_atol=error_control[1]) if error_control else {}
        hm = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
        om = {"bdf": O.METHOD_BDF, "tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
        s = H.Solver(m, p, nbatch=nb, sens=True, method=hm, **kw, **tol)
        y, sens, tot, mm = s.solve_dense_adaptive_sens(te, group=1, want_member_stats=True)
        yo, so, sto, failed = O.solve_dense_independent_sens(mid, np.asarray(p, dtype=float), te, nthreads=8, group=1, method=om, **kw, **tol)
        assert failed == 0 and tot["failed_members"] == 0 and (mm["status"] == 0).all()
        assert np.array_equal(mm["stats"].T, sto), "counters differ"
        assert np.array_equal(y, np.transpose(yo, (1, 0, 2))), "states differ"
        assert np.array_equal(sens, np.transpose(so, (0, 2, 1, 3))), "sensitivities differ"
        assert np.abs(sens).max() > 0
    # lock-step groups and models with root functions have no such kernel: refused, not silently routed elsewhere
    with pytest.raises(Exception):
        s.solve_dense_adaptive_sens(te, group=64)


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
@pytest.mark.parametrize("error_control", [None, (1e-6, [1e-7])])
def test_workgroup_per_member_bdf_with_sensitivities(H, O, det_pow, error_control, method):
    """Forward sensitivities in the workgroup-per-member forms (k_bdf_team_member<.., SENS>, k_sdirk_wave_member<.., SENS, TW>; dense run-time-compiled models with
    64 < n <= 140): the wavefront-per-member kernels' sensitivity code on the factors held in LDS.  Forty coupled oscillators (n = 80, three parameters) and a dense linear system with 100 states: every
    counter and every bit of states and sensitivities equal the oracle's solve_dense_sensitivities per member on the host twin."""
    import diffsl_models as D
    from diffsol_amd import diffsl as fe
    from diffsol_amd import _ffi
    rng = np.random.default_rng(18)
    nb = 40
    dense = ("in = [k]\nk { 1.0 }\nA_ij { (0:100, 0:100): -0.005, (0..100, 0..100): -1.0 }\nu_i { (0:100): 1.0 }\nlin_i { A_ij * u_j }\nF_i { k * lin_i + 0.1 * k * k }\n")
    cases = [(D.oscillators(40), np.stack([rng.uniform(20, 60, nb), rng.uniform(0.5, 2.0, nb), rng.uniform(0.005, 0.02, nb)], axis=1), [0.02, 0.1], dict(rtol=1e-5, atol=[1e-7])),
             (dense, rng.uniform(0.5, 2.0, (nb, 1)), [0.1, 0.7, 2.0], dict(rtol=1e-7, atol=[1e-9]))]
    L = _ffi.load_device_lib()
    for code, p, te, tol in cases:
        m, mid = fe.DiffslModel(code), D.host_model(O, code)
        assert m.form == fe.FORM_DYNAMIC and m.lane_model_id is None and L.dsh_model_has_wave_member_sens(m.model_id, 0) == 2
        kw = dict(sens_rtol=error_control[0], sens_atol=error_control[1]) if error_control else {}
        hm = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
        om = {"bdf": O.METHOD_BDF, "tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
        s = H.Solver(m, p, nbatch=nb, sens=True, method=hm, **kw, **tol)
        y, sens, tot, mm = s.solve_dense_adaptive_sens(te, group=1, want_member_stats=True)
        yo, so, sto, failed = O.solve_dense_independent_sens(mid, np.asarray(p, dtype=float), te, nthreads=8, group=1, method=om, **kw, **tol)
        assert failed == 0 and tot["failed_members"] == 0 and (mm["status"] == 0).all()
        assert np.array_equal(mm["stats"].T, sto), "counters differ"
        assert np.array_equal(y, np.transpose(yo, (1, 0, 2))), "states differ"
        assert np.array_equal(sens, np.transpose(so, (0, 2, 1, 3))), "sensitivities differ"
        assert np.abs(sens).max() > 0
