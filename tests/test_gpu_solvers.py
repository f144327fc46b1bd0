"""GPU parity tests of the integrators (rows a7-a15 of SURVEY §8) — they read like the reference's own integrator tests
(crates/diffsol/src/ode_solver/bdf.rs:1729-2631, sdirk.rs:676-1026): build the problem, run `test_ode_solver`'s loop, compare with
the known answers; plus bit-level comparison with the CPU oracle on identical (lock-step) ensembles, and size-independent properties at
BASELINE.json's full ensemble size."""
import numpy as np
import pytest

from helpers import METHOD, ORACLE_MODEL, robertson_params, times_of, weighted_error_norm

pytestmark = pytest.mark.gpu

ROB = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
DECADES = [0.0] + [0.4 * 10 ** k for k in range(0, 12)]


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


ALL_CASES = [("bdf_snapshots", n) for n in ["bdf_test_nalgebra_exponential_decay", "test_bdf_nalgebra_exponential_decay_algebraic", "test_bdf_nalgebra_robertson",
                                              "test_bdf_nalgebra_robertson_ode", "test_bdf_nalgebra_dydt_y2", "test_bdf_nalgebra_gaussian_decay"]] + \
            [("sdirk_snapshots", n) for n in ["test_tr_bdf2_nalgebra_exponential_decay2", "test_esdirk34_nalgebra_exponential_decay",
                                                "test_esdirk34_nalgebra_exponential_decay_algebraic", "test_tr_bdf2_nalgebra_robertson",
                                                "test_esdirk34_nalgebra_robertson", "test_tr_bdf2_nalgebra_robertson_ode"]]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("group,name", ALL_CASES)
def test_gpu_reproduces_reference_snapshot_counters_and_oracle_states(H, O, kats, group, name, fused):
    """The HIP path follows the reference's step sequence exactly: all 13 insta-snapshot counters match, and every interpolated output
    equals the CPU oracle's bit for bit (single IVP, nbatch = 1)."""
    spec = dict(kats["snapshot_problems"][name])
    if spec["model"] == "robertson_ode":
        spec["atol"] = list(np.tile(spec["atol"], spec.get("size", 1)))
    t = times_of(kats, spec["t"])
    kw = dict(model_size=spec.get("size", 0), rtol=spec["rtol"], atol=spec["atol"], h0=spec["h0"], method=METHOD[spec["method"]])
    s = H.Solver(spec["model"], spec["p"], fused=fused, **kw)
    y, _ = s.solve_to_points(t)
    assert s.stats() == kats[group][name]
    o = O.OracleSolver(ORACLE_MODEL[spec["model"]], spec["p"], **kw)
    yo, _ = o.solve_to_points(t)
    assert np.array_equal(y, yo)
    has_fused = spec["model"] in ("exponential_decay", "exponential_decay_with_algebraic", "robertson") or (spec["model"] == "robertson_ode" and spec.get("size", 1) == 1)
    assert s.fused == (fused and has_fused)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("method", ["bdf", "tr_bdf2"])
def test_stop_time_inside_the_first_step_keeps_the_factors_of_the_initial_step_size(H, O, method, fused):
    """A final time closer than the initial step: set_stop_time shortens h (and the operator's c) BEFORE the first step is taken.  The reference made
    its first factorisation of M - c0 J in the constructor (bdf.rs:289-293) and keeps it; the host integrator here defers that factorisation to
    its first step (the containers are allocated lazily) and must make it with the constructor's c0 all the same — Newton iterates, counters and bits
    against the oracle, which does not defer."""
    p = robertson_params(5)
    kw = dict(nbatch=5, model_size=1, method=METHOD[method], h0=1e-2, **ROB)
    for t_end in (1e-5, 3e-3):  # both inside [t0, t0 + h0]
        s = H.Solver("robertson_ode", p, fused=fused, **kw)
        o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, **kw)
        y, ncols, _ = s.solve(t_end)  # OdeSolverMethod::solve: set_stop_time(final_time) first (method.rs:227-258)
        yo, ncols_o = o.solve(t_end)
        assert np.array_equal(y, yo) and ncols == ncols_o and s.stats() == o.stats()


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("nb", [2, 67, 1000])
def test_lockstep_ensemble_is_bit_identical_to_oracle(H, O, nb, fused, method):
    """Same algorithm, same (max-over-batch) step sequence => identical bits for every member, state, derivative and difference array."""
    p = robertson_params(nb)
    kw = dict(nbatch=nb, model_size=1, method=METHOD[method], **ROB)
    s = H.Solver("robertson_ode", p, fused=fused, **kw)
    o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, **kw)
    pts = DECADES[:8]
    y, _ = s.solve_to_points(pts)
    yo, _ = o.solve_to_points(pts)
    assert np.array_equal(y, yo)
    assert s.stats() == o.stats()
    st, so = s.state(), o.state()
    assert st["t"] == so["t"] and st["h"] == so["h"] and st["order"] == so["order"]
    assert np.array_equal(st["y"], so["y"]) and np.array_equal(st["dy"], so["dy"])
    if method == "bdf":
        k = so["order"] + 3
        assert np.array_equal(s.diff()[:, :k], o.diff()[:, :k])
    # Robertson conserves y1+y2+y3 exactly in exact arithmetic; BDF/SDIRK preserve linear invariants up to the Newton tolerance
    assert np.abs(y.sum(-1) - 1.0).max() < 1e-6


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2"])
def test_reference_cuda_batched_exponential_decay(H, O, method):
    """test_bdf_cuda_exponential_decay_batched / test_tr_bdf2_cuda_exponential_decay_batched (bdf.rs:2497-2503, sdirk.rs:1011-1018)."""
    nb = 2
    p = [[0.1 * (b + 1), float(b + 1)] for b in range(nb)]
    s = H.Solver("exponential_decay", p, nbatch=nb, h0=1.0, method=METHOD[method])
    t = np.arange(10.0)
    y, _ = s.solve_to_points(t)
    for k in range(10):
        for b in range(nb):
            assert weighted_error_norm(y[k, b], np.full(2, (b + 1) * np.exp(-0.1 * (b + 1) * t[k])), [1e-6], 1e-6) < 20.0
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay"], p, nbatch=nb, h0=1.0, method=METHOD[method])
    assert np.array_equal(y, o.solve_to_points(t)[0])


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2"])
def test_reference_cuda_batched_exponential_decay_with_algebraic(H, O, method):
    """test_bdf_cuda_exponential_decay_with_algebraic_batched (bdf.rs:2625-2631, sdirk.rs:1020-1026): singular mass, nbatch 2."""
    nb = 2
    p = [[0.1 * (b + 1)] for b in range(nb)]
    s = H.Solver("exponential_decay_with_algebraic_batched", p, nbatch=nb, method=METHOD[method])
    t = np.arange(10.0) / 10
    y, _ = s.solve_to_points(t)
    for k in range(10):
        for b in range(nb):
            assert weighted_error_norm(y[k, b], np.full(3, np.exp(-0.1 * (b + 1) * t[k])), [1e-6], 1e-6) < 20.0
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay_with_algebraic_batched"], p, nbatch=nb, method=METHOD[method])
    assert np.array_equal(y, o.solve_to_points(t)[0])


def test_inconsistent_dae_initial_condition_is_made_consistent_on_device(H, O):
    """exponential_decay_with_algebraic starts at (1,1,0): InitOp + Newton with backtracking line search (state.rs:84-162) must give z=1."""
    s = H.Solver("exponential_decay_with_algebraic", [[0.1], [0.3], [0.2]], nbatch=3)
    st = s.state()
    assert np.allclose(st["y"], 1.0, atol=1e-9) and np.all(st["dy"][:, 2] == 0.0)
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay_with_algebraic"], [[0.1], [0.3], [0.2]], nbatch=3)
    assert np.array_equal(st["y"], o.state()["y"]) and st["h"] == o.state()["h"]


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2"])
def test_solve_and_tstop_match_oracle(H, O, method):
    """OdeSolverMethod::solve (method.rs:227-258): last time is exactly final_time, one column per accepted step."""
    nb = 5
    p = robertson_params(nb)
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, method=METHOD[method], **ROB)
    y, ncols, reason, ts, ys = s.solve(40.0, keep_trajectory=True)
    o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, nbatch=nb, model_size=1, method=METHOD[method], **ROB)
    yo, ncols_o = o.solve(40.0)
    assert reason == 2 and ncols == ncols_o and np.array_equal(y, yo)
    assert ts[0] == 0.0 and abs(ts[-1] - 40.0) <= 100 * 2.3e-16 * 80 and np.all(np.diff(ts) > 0)
    assert ys.shape == (ncols, nb, 3) and np.array_equal(ys[-1], y) and np.array_equal(ys[0], np.tile([1.0, 0.0, 0.0], (nb, 1)))


def test_solve_dense_interpolates_at_t_eval(H, O):
    nb = 4
    p = robertson_params(nb)
    t_eval = [0.0, 0.4, 4.0, 40.0]
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **ROB)
    y, reason = s.solve_dense(t_eval)
    assert reason == 2 and y.shape == (4, nb, 3)
    o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, nbatch=nb, model_size=1, **ROB)
    o.set_stop_time(t_eval[-1])
    col, out = 0, np.empty_like(y)
    while True:
        r = o.step()
        while col < len(t_eval) and t_eval[col] <= o.state()["t"]:
            out[col] = o.interpolate(t_eval[col]); col += 1
        if r == 2:
            break
    assert np.array_equal(y, out)


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2"])
def test_root_finder_stops_at_crossing(H, method):
    """exponential_decay_problem_with_root (g = y0 - 0.6): solve stops at t = -ln(0.6)/k (method.rs test_solve_stops_on_root)."""
    s = H.Solver("exponential_decay_with_root", [[0.1, 1.0], [0.1, 1.0]], nbatch=2, h0=1.0, method=METHOD[method])
    y, ncols, reason = s.solve(10.0)
    t_root, idx = s.root_info()
    assert reason == 1 and idx == 0 and abs(t_root - (-np.log(0.6) / 0.1)) < 1e-3
    assert weighted_error_norm(y[0], [0.6, 0.6], [1e-6], 1e-6) < 15.0
    # batch members that disagree on the crossing raise instead of silently following member 0 (the reference panics)
    s2 = H.Solver("exponential_decay_with_root", [[0.1, 1.0], [0.5, 1.0]], nbatch=2, h0=1.0, method=METHOD[method])
    with pytest.raises(H.DiffsolHipError):
        s2.solve(10.0)


def test_heat1d_trait_path_matches_oracle_and_fourier_series(H, O):
    """heat1d known-answer test (test_models/heat1d.rs:62-96) through the generic (run-time n) LU kernels."""
    mgrid = 10
    n, h = mgrid + 1, 1.0 / (mgrid + 2)
    times = [0.5 + 0.01 * i for i in range(5)]
    s = H.Solver("heat1d", [[1.0], [2.0]], nbatch=2, model_size=n, rtol=1e-6, atol=[1e-6])
    assert not s.fused
    y, _ = s.solve_to_points(times)
    o = O.OracleSolver(ORACLE_MODEL["heat1d"], [[1.0], [2.0]], nbatch=2, model_size=n, rtol=1e-6, atol=[1e-6])
    assert np.array_equal(y, o.solve_to_points(times)[0])
    x = (np.arange(n) + 1) * h
    for b, D in enumerate([1.0, 2.0]):
        for k, t in enumerate(times):
            ref = sum(np.sin((2 * m - 1) * np.pi * x) * np.exp(-(2 * m - 1) ** 2 * np.pi ** 2 * D * t) / (2 * m - 1) ** 2 for m in range(1, 100)) * 8 / np.pi ** 2
            assert weighted_error_norm(y[k, b], ref, [1e-4], 1e-4) < 20.0


def test_rlc_dae_matches_oracle_bitwise(H, O):
    """Electrical-circuits DAE (examples/electrical-circuits), ESDIRK34 with a singular mass matrix.  The source term's sin() is
    include/diffsol_detpow.h's on both sides, so this model is bit-identical too."""
    p = [[100.0, 1.0, 1e-3, 10.0, 100.0, 0.05], [150.0, 1.0, 2e-3, 10.0, 100.0, 0.05]]
    s = H.Solver("rlc", p, nbatch=2, method=METHOD["esdirk34"])
    o = O.OracleSolver(ORACLE_MODEL["rlc"], p, nbatch=2, method=METHOD["esdirk34"])
    y, _, _ = s.solve(0.05)
    yo, _ = o.solve(0.05)
    assert np.array_equal(y, yo) and s.stats() == o.stats()


def test_spm_battery_ensemble_matches_oracle_bitwise_and_stops_at_the_voltage_cutoff(H, O):
    """Single-particle battery model (book/src/primer/src/spm.ds, n=42): the right-hand side is linear, so the lock-step ensemble is
    bit-identical to the oracle; so is the stop condition V < 3.105 (tanh / asinh / exp from include/diffsol_detpow.h on both sides): same root
    time to the last bit."""
    cur = np.linspace(0.6, 1.4, 9)[:, None]
    s = H.Solver("spm", cur, nbatch=9, model_size=20, rtol=1e-6, atol=[1e-6])
    assert s.n == 42 and not s.fused
    o = O.OracleSolver(ORACLE_MODEL["spm"], cur, nbatch=9, model_size=20, rtol=1e-6, atol=[1e-6])
    times = [60.0, 300.0, 900.0]
    y, _ = s.solve_to_points(times)
    yo, _ = o.solve_to_points(times)
    assert np.array_equal(y, yo)
    assert np.allclose(y[-1][:, 0], cur[:, 0] * 900.0 / 3600.0, rtol=1e-6)  # discharge capacity [Ah] = I t
    same = np.full((4, 1), 1.0)
    s2 = H.Solver("spm", same, nbatch=4, model_size=20, rtol=1e-6, atol=[1e-6])
    o2 = O.OracleSolver(ORACLE_MODEL["spm"], same, nbatch=4, model_size=20, rtol=1e-6, atol=[1e-6])
    _, _, reason = s2.solve(3600.0)
    o2.solve(3600.0)
    (t_root, idx), (t_ref, idx_ref) = s2.root_info(), o2.root_info()
    assert reason == 1 and idx == idx_ref == 0 and t_root == t_ref and 2000.0 < t_root < 3000.0


@pytest.mark.parametrize("env", [{"DSH_FUSE_ACCEPT": "1"}, {"DSH_NEWTON_NIT": "1"}, {"DSH_NEWTON_NIT": "4", "DSH_FUSE_ACCEPT": "1"}, {"DSH_NEWTON_PIPELINE": "0"},
                                 {"DSH_SYNC_MODE": "sync"}])
def test_launch_structure_knobs_do_not_change_a_single_bit(H, O, monkeypatch, env):
    """Iterations per Newton launch, cross-step prelaunch, the fused accept+Newton launch and polling vs stream synchronisation only change how
    the work is cut into launches: states, difference array and all counters stay bit-identical to the oracle's lock-step run."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    nb = 300
    p = robertson_params(nb)
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **ROB)
    o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, nbatch=nb, model_size=1, **ROB)
    times = [0.4, 4.0, 400.0, 4e4]
    y, _ = s.solve_to_points(times)
    yo, _ = o.solve_to_points(times)
    assert np.array_equal(y, yo) and np.array_equal(s.diff(), o.diff())
    so, ss = o.stats(), s.stats()
    assert all(ss[k] == so[k] for k in ("number_of_steps", "number_of_nonlinear_solver_iterations", "number_of_linear_solver_setups",
                                        "number_of_error_test_failures", "number_of_nonlinear_solver_fails"))


# ------------------------------------------------------------------ BASELINE.json config 2 at full size: size-independent properties
@pytest.fixture(scope="module", params=["default", "host_lockstep"])
def full_size_run(H, request):
    """`default`: dshs_solve_dense as a user gets it (DSHS_ENSEMBLE_AUTO -> the device-resident integrator, wavefront groups of 64 — bench.py's
    `value` path); `host_lockstep`: the trait-boundary path, one (t, h, order) for all 100 000 members."""
    nb = 100_000
    p = robertson_params(nb)
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, ensemble_mode=0 if request.param == "host_lockstep" else None, **ROB)
    assert s.ensemble_mode() == ((0, 0) if request.param == "host_lockstep" else (-1, 64))
    t_eval = [0.4, 4.0, 40.0, 400.0, 4e3, 4e4, 4e5]
    y, reason = s.solve_dense(t_eval)
    mode, tot = s.last_solve_info()
    assert mode == (0 if request.param == "host_lockstep" else 64) and tot["failed_members"] == 0
    st = {k: v // nb for k, v in tot.items()}  # mean per member
    if request.param == "host_lockstep":
        assert st["number_of_steps"] == s.stats()["number_of_steps"]
    return p, t_eval, y, st, reason


def test_default_solve_dense_is_the_device_resident_integrator_and_equals_the_oracle_group_by_group(H, O, request):
    """The drop-in default: dshs_solve_dense routes to dsh_bdf_solve_adaptive in wavefront lock-step groups of 64, i.e. the
    reference's batched semantics with nbatch = 64 per group: every member equals the oracle's lock-step batched run of its own group bit for bit
    (ragged last group included); an ensemble of <= 64 members is ONE group = the host-driven lock-step ensemble, bit for bit."""
    nb = 64 * 5 + 23
    p = robertson_params(nb, seed=4242)
    t_eval = [0.4, 4.0, 40.0, 400.0, 4e3]
    O.set_det_pow(True)           # the device-resident integrators' pow is include/diffsol_detpow.h: same switch on the oracle ...
    H.set_deterministic_pow(True)  # ... and on the host-driven integrators (libm by default = the reference's arithmetic)
    request.addfinalizer(lambda: (O.set_det_pow(False), H.set_deterministic_pow(False)))
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **ROB)
    assert s.ensemble_mode() == (-1, 64)
    y, reason = s.solve_dense(t_eval)
    mode, tot = s.last_solve_info()
    assert reason == 2 and mode == 64 and tot["failed_members"] == 0
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], p, t_eval, model_size=1, group=64, **ROB)
    assert failed == 0 and np.array_equal(y, np.transpose(yo, (1, 0, 2)))
    assert tot["number_of_steps"] == int(so[:, 0].sum()) and tot["number_of_nonlinear_solver_iterations"] == int(so[:, 1].sum())
    # the solver object was not advanced: calling again gives the same bits; the host-driven mode continues to work on the same object
    y2, _ = s.solve_dense(t_eval)
    assert np.array_equal(y2, y)
    small = H.Solver("robertson_ode", p[:40], nbatch=40, model_size=1, **ROB)
    ya, _ = small.solve_dense(t_eval)
    host = H.Solver("robertson_ode", p[:40], nbatch=40, model_size=1, ensemble_mode=0, **ROB)
    yh, _ = host.solve_dense(t_eval)
    assert small.last_solve_info()[0] == 64 and host.last_solve_info()[0] == 0 and np.array_equal(ya, yh)
    assert small.last_solve_info()[1]["number_of_steps"] == host.last_solve_info()[1]["number_of_steps"] == 40 * host.stats()["number_of_steps"]
    # a solver that was stepped by hand continues on the host path (the device-resident integrators start from t0)
    host2 = H.Solver("robertson_ode", p[:40], nbatch=40, model_size=1, **ROB)
    host2.step()
    assert host2.ensemble_mode() == (-1, 0)


def test_default_solve_dense_fails_like_the_reference_when_a_member_fails(H):
    """solve_dense returns Err when the integration fails; the ensemble default does the same when ANY member fails (per-member status is what
    dshs_solve_dense_adaptive is for)."""
    p = robertson_params(70, seed=3)
    s = H.Solver("robertson_ode", p, nbatch=70, model_size=1, options=dict(max_error_test_failures=1, max_nonlinear_solver_failures=1), **ROB)
    with pytest.raises(H.DiffsolHipError) as e:
        s.solve_dense([0.4, 4e5])
    assert "ensemble members failed" in str(e.value) and e.value.code <= -100


def test_full_size_ensemble_invariants(full_size_run):
    p, t_eval, y, st, reason = full_size_run
    assert reason == 2 and y.shape == (7, 100_000, 3) and np.isfinite(y).all()
    assert np.abs(y.sum(-1) - 1.0).max() < 1e-7          # linear invariant y1+y2+y3 = 1
    assert y.min() > -1e-6                               # concentrations stay non-negative up to atol
    assert np.all(np.diff(y[:, :, 0], axis=0) < 1e-9)    # y1 decays monotonically, y3 grows
    assert np.all(np.diff(y[:, :, 2], axis=0) > -1e-9)
    assert st["number_of_steps"] > 100 and st["number_of_nonlinear_solver_iterations"] >= st["number_of_steps"]


def test_first_and_last_wavefront_group_of_the_full_size_run_equal_the_oracle_bitwise(O, full_size_run, request):
    """BASELINE config 2 at its full 100 000 members, bit for bit where it costs nothing: the first wavefront group (members 0..63) and the
    last, ragged one (members 99 968..99 999: 32 live lanes next to 32 shadow lanes) of the default run against the oracle's lock-step batched run of
    exactly those members — the device-resident integrator's result for a member does not depend on how many other groups the launch has."""
    p, t_eval, y, _, _ = full_size_run
    if request.node.callspec.params.get("full_size_run") == "host_lockstep":
        pytest.skip("the host-driven lock-step ensemble has ONE step sequence for all 100 000 members: its groups are not independent problems")
    O.set_det_pow(True)
    request.addfinalizer(lambda: O.set_det_pow(False))
    for lo, hi in ((0, 64), (99_968, 100_000)):
        yo, _, failed = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], p[lo:hi], t_eval, model_size=1, group=64, **ROB)
        assert failed == 0 and np.array_equal(y[:, lo:hi], np.transpose(yo, (1, 0, 2)))


def test_full_size_ensemble_members_agree_with_independent_cpu_solves(O, full_size_run):
    """'step counts may differ, solution error must not': a random sample of members, each re-solved on the CPU as an independent IVP
    with its own adaptive step sequence (the reference's CPU usage pattern), agrees within the reference's acceptance norm."""
    p, t_eval, y, _, _ = full_size_run
    rng = np.random.default_rng(1)
    for b in rng.choice(len(p), 48, replace=False):
        o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p[b], model_size=1, **ROB)
        yo, _ = o.solve_to_points(t_eval)
        for k in range(len(t_eval)):
            assert weighted_error_norm(y[k, b], yo[k, 0], ROB["atol"], ROB["rtol"]) < 20.0


def test_subensemble_of_full_run_is_bit_identical_when_it_shares_the_step_sequence(H, O):
    """Lock-step semantics: members only interact through the max-norm, so an ensemble and the oracle on the same 4096 members agree bitwise."""
    nb = 4096
    p = robertson_params(nb, seed=99)
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **ROB)
    o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, nbatch=nb, model_size=1, **ROB)
    y, _, _ = s.solve(4e3)
    yo, _ = o.solve(4e3)
    assert np.array_equal(y, yo) and s.stats() == o.stats()


def test_tight_tolerance_ensemble_within_1e6_relative_of_independent_cpu_reference(H, O):
    """north_star: 'solution within 1e-6 rel of CPU reference'.  At rtol 1e-10 the lock-step ensemble and independent per-IVP CPU solves
    (different step sequences) agree to 1e-6 relative on every component above its absolute tolerance."""
    nb = 256
    p = robertson_params(nb, seed=5)
    kw = dict(rtol=1e-10, atol=[1e-14, 1e-18, 1e-12])
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **kw)
    y, _, _ = s.solve(40.0)
    for b in range(0, nb, 8):
        yo, _ = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p[b], model_size=1, **kw).solve(40.0)
        assert np.max(np.abs(y[b] - yo[0]) / np.abs(yo[0])) < 1e-6


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
def test_after_a_root_stop_the_state_sits_at_the_root_with_interpolated_y_and_dy(H, O, method):
    """OdeSolverMethod::solve on RootFound calls state_mut_back(t_root) (method.rs, bdf.rs:1232-1262, runge_kutta.rs:396-434): state.y and state.dy
    become the step's interpolants at the root time (BDF difference polynomial and its derivative; TR-BDF2 beta polynomial; ESDIRK34 Hermite) —
    no extra right-hand-side evaluation, so the call counters stay the reference's."""
    p = [[0.1, 1.0], [0.1, 1.0], [0.1, 1.0]]
    s = H.Solver("exponential_decay_with_root", p, nbatch=3, method=METHOD[method], rtol=1e-6, atol=[1e-6, 1e-6])
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay_with_root"], p, nbatch=3, method=METHOD[method], rtol=1e-6, atol=[1e-6, 1e-6])
    _, _, reason = s.solve(50.0)
    o.solve(50.0)
    t_root = s.root_info()[0]
    assert reason == 1 and s.root_info() == o.root_info() and abs(t_root - np.log(1.0 / 0.6) / 0.1) < 1e-4 and s.stats() == o.stats()
    st = s.state()
    assert st["t"] == t_root and np.array_equal(st["y"], o.interpolate(t_root)) and np.array_equal(st["dy"], o.interpolate_dy(t_root))
    assert np.allclose(st["dy"], -0.1 * st["y"], rtol=1e-3)


@pytest.mark.parametrize("model,size,times", [("spm", 5, [30.0, 200.0]), ("heat1d", 16, [0.005, 0.03]), ("robertson_ode", 3, [0.4, 4.0, 40.0])])
def test_large_ensembles_of_run_time_sized_models_use_the_difference_array_kernels_with_the_same_bits(H, O, monkeypatch, model, size, times):
    """nbatch >= 8192, no fused Newton kernel: Bdf still runs rescale / predict / accept + order-selection norms as single launches (one lane per
    system).  Same states and counters as the oracle's lock-step run and as the 1:1 trait composition (DSH_FUSE_LA=0)."""
    nb = 8192
    rng = np.random.default_rng(11)
    if model == "robertson_ode":
        p, tol = robertson_params(nb, seed=11), dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6] * 3)
    else:
        p, tol = rng.uniform(0.6, 0.9, (nb, 1)), dict(rtol=1e-6, atol=[1e-6])  # (battery: well before any member's voltage cut-off)
    s = H.Solver(model, p, nbatch=nb, model_size=size, **tol)
    assert not s.fused
    y, _ = s.solve_to_points(times)
    o = O.OracleSolver(ORACLE_MODEL[model], p, nbatch=nb, model_size=size, **tol)
    yo, _ = o.solve_to_points(times)
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    monkeypatch.setenv("DSH_FUSE_LA", "0")
    s0 = H.Solver(model, p, nbatch=nb, model_size=size, **tol)
    y0, _ = s0.solve_to_points(times)
    assert np.array_equal(y0, y) and s0.stats() == s.stats()
