"""Device-resident integrators (dsh_bdf_solve_adaptive, dsh_sdirk_solve_resident, dsh_bdf_solve_wave_member) against the oracle run
the way diffsol's CPU path treats a parameter sweep: one independent IVP per member (oracle.solve_dense_independent), or one lock-step batched problem per
64-member group.

Parity statement.  The kernels repeat the oracle's arithmetic operation for operation; the only thing they cannot share with a CPU run is libm.
* Default mode (deterministic_pow=True): pow() is include/diffsol_detpow.h on the device, and the oracle is switched to the same function
  (orc_set_det_pow) — then EVERYTHING must agree bit for bit: the `*deterministic*` tests at the end of this file.  The oracle reproduces all reference
  snapshots in that mode too (tests/test_oracle_golden.py).
* ocml mode (deterministic_pow=False) against the oracle with libm's pow: both are accurate to <= 1 ulp but not identical.  A 1-ulp difference in h
  leaves every accept/reject decision unchanged unless a test value sits within rounding of its threshold, but it is fed back through the step-size
  controller over a few hundred steps, so the states of a member with IDENTICAL decisions still drift apart — measured up to 3e-6 relative at rtol
  1e-4.  The tests of that mode (run_pair below) require identical per-member counters for > 90-98 % of the members, states within rtol/10 where the
  counters agree and within a few rtol where they do not, and, at tight tolerances, all members within 1e-6 relative of the CPU result (north_star)."""
import numpy as np
import pytest

from helpers import ORACLE_MODEL
from bench import robertson_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]
ROB = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])


def run_pair(H, O, model, p, t_eval, model_size, group=1, method=0, **tol):
    nb = len(p)
    s = H.Solver(model, p, nbatch=nb, model_size=model_size, method=method, **tol)
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group, deterministic_pow=False)  # ocml pow vs the oracle's libm pow
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL[model], np.asarray(p, dtype=float), t_eval, model_size=model_size, nthreads=8, group=group,
                                               method=method, **tol)
    run_pair.member, run_pair.oracle_roots = m, O.solve_dense_independent.last_roots
    return y, tot, m["stats"], m["status"], np.transpose(yo, (1, 0, 2)), so, failed


def test_adaptive_robertson_members_match_independent_cpu_solves(H, O):
    nb = 1000
    p = robertson_params(nb)
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "robertson_ode", p, T_EVAL, 1, **ROB)
    assert failed == 0 and (status == 0).all() and tot["failed_members"] == 0
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.98, f"only {same.mean():.3f} of the members reproduce the CPU step sequence"
    assert tot["number_of_steps"] == int(stats[0].sum()) and tot["number_of_nonlinear_solver_iterations"] == int(stats[1].sum())
    # members with the same decisions: states well inside the tolerance (see module docstring)
    assert np.allclose(y[:, same], yo[:, same], rtol=1e-5, atol=1e-300)
    # all members: within the solver tolerance scale of the CPU result, mass conserved
    assert np.allclose(y, yo, rtol=5e-3, atol=1e-9)
    assert np.abs(y.sum(axis=2) - 1.0).max() < 1e-9
    # per-member control really is per member: step counts differ across the sweep and are far below the lock-step ensemble's
    assert stats[0].min() < stats[0].max()


def test_wavefront_lockstep_groups_match_the_batched_oracle_with_nbatch_64(H, O):
    """group=64: each wavefront integrates its 64 members in lock-step (max-norms over the wavefront) — the reference's batched semantics with
    nbatch = 64 per group, with no host in the loop.  Compared with the oracle's lock-step batched run of every group (the last group is ragged:
    1000 = 15 x 64 + 40).  Same libm caveat as the per-member mode."""
    nb = 1000
    p = robertson_params(nb)
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "robertson_ode", p, T_EVAL, 1, group=64, **ROB)
    assert failed == 0 and (status == 0).all() and tot["failed_members"] == 0
    for g in range(0, nb, 64):  # counters are per group
        assert (stats[:, g:g + 64] == stats[:, g:g + 1]).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9, f"only {same.mean():.3f} of the members sit in a group that reproduces the CPU step sequence"
    assert np.allclose(y[:, same], yo[:, same], rtol=1e-5, atol=1e-300)
    assert np.allclose(y, yo, rtol=5e-3, atol=1e-9)
    assert np.abs(y.sum(axis=2) - 1.0).max() < 1e-9


def test_adaptive_solution_within_1e6_relative_of_cpu_reference_at_tight_tolerance(H, O):
    """north_star: solution within 1e-6 rel of the CPU reference (step counts may differ)."""
    nb = 256
    p = robertson_params(nb, seed=7)
    tol = dict(rtol=1e-10, atol=[1e-14, 1e-18, 1e-12])
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "robertson_ode", p, T_EVAL[:5], 1, **tol)
    assert failed == 0 and (status == 0).all()
    rel = np.abs(y - yo) / np.maximum(np.abs(yo), 1e-30)
    assert rel.max() < 1e-6


def test_adaptive_exponential_decay_counters_and_analytic_solution(H, O):
    nb = 130
    k = 0.05 * (np.arange(nb) + 1)
    p = np.stack([k, np.arange(nb) + 1.0], axis=1)
    t_eval = [1.0, 2.5, 9.0]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "exponential_decay", p, t_eval, 0, rtol=1e-6, atol=[1e-6, 1e-6])
    assert failed == 0 and (status == 0).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.95
    assert np.allclose(y[:, same], yo[:, same], rtol=1e-7, atol=0)
    exact = p[None, :, 1:2] * np.exp(-p[None, :, 0:1] * np.asarray(t_eval)[:, None, None]) * np.ones((1, 1, 2))
    assert np.allclose(y, exact, rtol=2e-4, atol=2e-5)  # atol of the solve is 1e-6


def test_adaptive_rejects_unsupported_models_and_bad_t_eval(H):
    s3 = H.Solver("gaussian_decay", [[1.0] * 330], nbatch=1, model_size=330, method=1)  # run-time sized, dense Jacobian, n = 330: no lane-per-member form and too large for the wavefront- / workgroup-per-member kernels (n <= 320)
    with pytest.raises(H.DiffsolHipError) as e:
        s3.solve_dense_adaptive([0.1])
    assert e.value.code == -6
    s2 = H.Solver("robertson_ode", [[0.04, 1e4, 3e7]], nbatch=1, model_size=1, **ROB)
    with pytest.raises(H.DiffsolHipError):
        s2.solve_dense_adaptive([2.0, 1.0])


def test_adaptive_full_size_ensemble_invariants(H):
    """BASELINE.json configs[1] at full size through the one-launch path: finite, mass conserving, monotone species."""
    nb = 100_000
    p = robertson_params(nb)
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **ROB)
    y, tot = s.solve_dense_adaptive(T_EVAL)
    assert tot["failed_members"] == 0 and np.isfinite(y).all()
    assert np.abs(y.sum(axis=2) - 1.0).max() < 1e-9
    assert (np.diff(y[:, :, 0], axis=0) <= 1e-12).all() and (np.diff(y[:, :, 2], axis=0) >= -1e-12).all()
    assert 150 * nb < tot["number_of_steps"] < 400 * nb


def test_fast_arithmetic_variant_at_full_size_makes_the_step_decisions_of_the_exact_kernel(H, O):
    """tier-level parity evidence for the opt-in `deterministic_pow = 2` build (bench.py's `fast_variant` extra: contracted multiply-adds,
    reciprocal-math division, ocml pow) on BASELINE config 2 at its full 100 000 members in wavefront lock-step groups: every member's five counters (steps, Newton
    iterations, LU setups, error-test failures, Newton failures) equal the exact kernel's — the two builds take the same step-size, order and refactorisation decisions in
    all 1563 groups — and every state above its absolute tolerance is within 1e-9 relative of the exact kernel's, which is bit-identical to the oracle (first and last
    group checked here against the oracle, the whole tier in tests/test_gpu_solvers.py)."""
    nb = 100_000
    p = robertson_params(nb)
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **ROB)
    ye, tote, me = s.solve_dense_adaptive(T_EVAL, group=64, deterministic_pow=1, want_member_stats=True)
    yf, totf, mf = s.solve_dense_adaptive(T_EVAL, group=64, deterministic_pow=2, want_member_stats=True)
    assert tote["failed_members"] == 0 and totf["failed_members"] == 0
    assert np.array_equal(me["stats"], mf["stats"]) and np.array_equal(me["status"], mf["status"])
    assert totf["number_of_steps"] == tote["number_of_steps"] and totf["number_of_nonlinear_solver_iterations"] == tote["number_of_nonlinear_solver_iterations"]
    big = np.abs(ye) > np.asarray(ROB["atol"])[None, None, :]
    assert (np.abs(yf - ye)[big] / np.abs(ye)[big]).max() < 1e-9
    assert not np.array_equal(yf, ye)  # it IS another arithmetic
    O.set_det_pow(True)
    try:
        for lo, hi in ((0, 64), (99_968, 100_000)):
            yo, _, failed = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], p[lo:hi], T_EVAL, model_size=1, group=64, **ROB)
            assert failed == 0 and np.array_equal(ye[:, lo:hi], np.transpose(yo, (1, 0, 2)))
            bo = np.abs(np.transpose(yo, (1, 0, 2))) > np.asarray(ROB["atol"])[None, None, :]
            assert (np.abs(yf[:, lo:hi] - np.transpose(yo, (1, 0, 2)))[bo] / np.abs(np.transpose(yo, (1, 0, 2)))[bo]).max() < 1e-9
    finally:
        O.set_det_pow(False)


def test_the_library_default_arithmetic_of_solve_dense_is_the_fast_build_and_exact_on_request(H, O):
    """Round 6: Solver.solve_dense (dshs_solve_dense) in its device-resident modes launches the FAST-arithmetic BDF by default where that build exists
    (dshs_set_resident_arithmetic / DSH_RESIDENT_ARITH; this tier pins `exact` in conftest.py).  ARITH_FAST gives the bits of solve_dense_adaptive(deterministic_pow=2),
    ARITH_EXACT those of deterministic_pow=1 (= the oracle's); models without a fast build (a run-time-compiled banded model here) run the exact kernel under either setting."""
    assert H.get_resident_arithmetic() == H.ARITH_EXACT  # conftest.py
    nb = 640
    p = robertson_params(nb, seed=11)
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, ensemble_mode=H.ENSEMBLE_WAVEFRONT, **ROB)
    ye, _ = s.solve_dense_adaptive(T_EVAL, group=64, deterministic_pow=1)
    yf, _ = s.solve_dense_adaptive(T_EVAL, group=64, deterministic_pow=2)
    try:
        y_exact = s.solve_dense(T_EVAL)[0]
        H.set_resident_arithmetic(H.ARITH_FAST)
        y_fast = s.solve_dense(T_EVAL)[0]
        # a model without a fast build: the banded lane-per-member form of heat1d (run-time compiled)
        s2 = H.Solver("heat1d", np.linspace(0.5, 2.0, 64)[:, None], nbatch=64, model_size=20, rtol=1e-6, atol=[1e-6], ensemble_mode=H.ENSEMBLE_PER_MEMBER)
        y_sd_fast = s2.solve_dense([0.01, 0.1])[0]
        H.set_resident_arithmetic(H.ARITH_EXACT)
        y_sd_exact = s2.solve_dense([0.01, 0.1])[0]
    finally:
        H.set_resident_arithmetic(H.ARITH_EXACT)
    assert np.array_equal(y_exact, ye) and np.array_equal(y_fast, yf) and not np.array_equal(yf, ye)
    assert np.array_equal(y_sd_fast, y_sd_exact)
    big = np.abs(ye) > np.asarray(ROB["atol"])[None, None, :]
    assert (np.abs(yf - ye)[big] / np.abs(ye)[big]).max() < 1e-9


def test_the_default_arithmetic_stays_within_1e6_relative_of_independent_cpu_solves_at_tight_tolerances(H, O):
    """north_star's bar for the path bench.py times (library default arithmetic, wavefront lock-step groups): at tolerances where the integration error is below it,
    every state within 1e-6 relative of the oracle's independent libm-pow solves of the same groups."""
    p = robertson_params(256, seed=4)
    tight = dict(rtol=1e-9, atol=[1e-13, 1e-17, 1e-11])
    s = H.Solver("robertson_ode", p, nbatch=len(p), model_size=1, ensemble_mode=H.ENSEMBLE_WAVEFRONT, **tight)
    try:
        H.set_resident_arithmetic(H.ARITH_FAST)
        y = s.solve_dense(T_EVAL[:5])[0]
    finally:
        H.set_resident_arithmetic(H.ARITH_EXACT)
    yo, _, failed = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], np.asarray(p, dtype=float), T_EVAL[:5], model_size=1, nthreads=8, group=64, **tight)
    yo = np.transpose(yo, (1, 0, 2))
    big = np.abs(yo) > 1e-7
    assert failed == 0 and (np.abs(y - yo)[big] / np.abs(yo)[big]).max() < 1e-6


# ------------------------------------------------------------------ device-resident TR-BDF2 / ESDIRK34 (dsh_sdirk_solve_resident)
@pytest.mark.parametrize("group", [1, 64])
@pytest.mark.parametrize("method", [1, 2])
def test_resident_sdirk_exponential_decay_matches_oracle(H, O, method, group):
    nb = 130
    k = 0.05 * (np.arange(nb) + 1)
    p = np.stack([k, np.arange(nb) + 1.0], axis=1)
    t_eval = [1.0, 2.5, 9.0]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "exponential_decay", p, t_eval, 0, group=group, method=method, rtol=1e-6, atol=[1e-6, 1e-6])
    assert failed == 0 and (status == 0).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9
    assert np.allclose(y[:, same], yo[:, same], rtol=1e-7, atol=0)
    exact = p[None, :, 1:2] * np.exp(-p[None, :, 0:1] * np.asarray(t_eval)[:, None, None]) * np.ones((1, 1, 2))
    assert np.allclose(y, exact, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("group", [1, 64])
def test_resident_bdf_dae_with_consistent_initialisation(H, O, group):
    """The BDF kernel on mass-matrix models: Robertson DAE (SUNDIALS reference problem) and the inconsistent algebraic exponential decay."""
    p = robertson_params(70)
    tol = dict(rtol=1e-4, atol=[1e-8, 1e-6, 1e-6])
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "robertson", p, T_EVAL[:5], 0, group=group, method=0, **tol)
    assert failed == 0 and (status == 0).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9 and np.allclose(y[:, same], yo[:, same], rtol=1e-5, atol=1e-300) and np.allclose(y, yo, rtol=5e-3, atol=1e-9)
    assert np.abs(y.sum(axis=2) - 1.0).max() < 1e-7
    pk = (0.1 * (np.arange(40) + 1))[:, None]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "exponential_decay_with_algebraic", pk, [1.0, 5.0], 0, group=group, method=0, rtol=1e-6,
                                                     atol=[1e-6] * 3)
    assert failed == 0 and (status == 0).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9 and np.allclose(y[:, same], yo[:, same], rtol=1e-7, atol=1e-12)
    assert np.allclose(y[..., 2], y[..., 1], atol=1e-9)


@pytest.mark.parametrize("group", [1, 64])
def test_resident_sdirk_dae_with_consistent_initialisation(H, O, group):
    """Mass-matrix models: Robertson DAE (consistent initial values) and exponential decay with an algebraic equation whose initial value is
    INCONSISTENT (y = (1, 1, 0) but 0 = y2 - y1): the device runs InitOp's Newton with the backtracking line search per member / per group."""
    p = robertson_params(70)
    tol = dict(rtol=1e-4, atol=[1e-8, 1e-6, 1e-6])
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "robertson", p, [0.4, 4.0, 40.0], 0, group=group, method=1, **tol)
    assert failed == 0 and (status == 0).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9 and np.allclose(y[:, same], yo[:, same], rtol=1e-5, atol=1e-300) and np.allclose(y, yo, rtol=5e-3, atol=1e-9)
    assert np.abs(y.sum(axis=2) - 1.0).max() < 1e-7  # the algebraic constraint
    pk = (0.1 * (np.arange(40) + 1))[:, None]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "exponential_decay_with_algebraic", pk, [1.0, 5.0], 0, group=group, method=2, rtol=1e-6,
                                                     atol=[1e-6] * 3)
    assert failed == 0 and (status == 0).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9 and np.allclose(y[:, same], yo[:, same], rtol=1e-7, atol=1e-12)
    assert np.allclose(y[..., 2], y[..., 1], atol=1e-9) and np.allclose(y[..., 0], np.exp(-pk[None, :, 0] * np.asarray([1.0, 5.0])[:, None]), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_resident_sdirk_per_member_events(H, O, method):
    """BASELINE config 5's point: every member stops at ITS OWN event.  exponential_decay_with_root (g = y0 - 0.6) with a different decay rate per
    member: root times -ln(0.6)/k_b differ by 20x across the ensemble; the lock-step backend refuses this (batch mismatch), the per-member kernel
    returns each member's root time, root index, number of valid columns and the state at the root."""
    nb = 100
    k = 0.05 * (np.arange(nb) + 1)
    p = np.stack([k, np.ones(nb)], axis=1)
    t_eval = [0.5, 1.0, 2.0, 4.0, 8.0, 16.0]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "exponential_decay_with_root", p, t_eval, 0, group=1, method=method, rtol=1e-6, atol=[1e-6, 1e-6])
    m, ref = run_pair.member, run_pair.oracle_roots
    assert failed == 0 and (status == 0).all()
    t_exact = -np.log(0.6) / k
    hit = t_exact < t_eval[-1]
    assert (m["root_idx"][hit] == 0).all() and (m["root_idx"][~hit] == -1).all()
    assert np.allclose(m["t_root"][hit], t_exact[hit], rtol=2e-4)
    assert np.array_equal(m["root_idx"], ref["root_idx"]) and np.array_equal(m["ncols"], ref["ncols"])
    assert np.allclose(m["t_root"][hit], ref["t_root"][hit], rtol=1e-7)
    for b in np.flatnonzero(hit):
        nc = m["ncols"][b]
        assert abs(y[nc - 1, b, 0] - 0.6) < 1e-5 and np.isnan(y[nc:, b]).all()  # last valid column = state at the root
    ok = np.isfinite(yo)
    assert np.array_equal(np.isfinite(y), ok) and np.allclose(y[ok], yo[ok], rtol=1e-6, atol=1e-12)
    # the wavefront lock-step mode reports the disagreement instead of following member 0
    s = H.Solver("exponential_decay_with_root", p, nbatch=nb, method=method, rtol=1e-6, atol=[1e-6, 1e-6])
    _, tot64, m64 = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=64)
    assert (m64["status"] == 20).all() and tot64["failed_members"] == nb


def test_resident_esdirk34_rlc_config5_members_track_independent_cpu_solves(H, O):
    """Config 5 (RLC DAE, ESDIRK34, root iR - i_thresh) at reduced size, per-member control: every member against its own CPU solve (ocml pow on
    the device, libm pow in the oracle: tolerance; the bitwise statement is in the deterministic tests below)."""
    nb = 200
    rng = np.random.default_rng(5)
    R, Cc = rng.uniform(50.0, 200.0, nb), np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
    p = np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, 0.03)], axis=1)
    t_eval = [0.002, 0.005, 0.01, 0.02, 0.05]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "rlc", p, t_eval, 1, group=1, method=2, rtol=1e-6, atol=[1e-6] * 4)
    m, ref = run_pair.member, run_pair.oracle_roots
    assert failed == 0 and (status == 0).all()
    assert np.array_equal(m["root_idx"], ref["root_idx"]) and np.array_equal(m["ncols"], ref["ncols"])
    hit = m["root_idx"] >= 0
    assert 0 < hit.sum() < nb  # some members reach the threshold current, at different times; others never do
    assert np.allclose(m["t_root"][hit], ref["t_root"][hit], rtol=1e-6) and np.ptp(m["t_root"][hit]) > 1e-3
    ok = np.isfinite(yo)
    assert np.array_equal(np.isfinite(y), ok) and np.allclose(y[ok], yo[ok], rtol=1e-5, atol=1e-9)
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9


# ------------------------------------------------------------------ one wavefront per member: run-time-sized models, n <= 64 (dsh_bdf_solve_wave_member)
def test_wave_member_spm_config4_members_match_cpu_solves_and_stop_at_their_own_cutoff(H, O):
    """BASELINE config 4 at reduced size: the single-particle battery model (n = 42), one member per wavefront, each with its own step sizes and its
    own voltage cut-off event.  The right-hand side is linear, so without events states match the CPU solves to rounding of h (pow is ocml's)."""
    nb = 40
    cur = np.linspace(0.6, 1.4, nb)[:, None]
    t_eval = [60.0, 600.0, 1200.0]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "spm", cur, t_eval, 20, group=1, method=0, rtol=1e-6, atol=[1e-6])
    m, ref = run_pair.member, run_pair.oracle_roots
    assert failed == 0 and (status == 0).all() and (m["root_idx"] == -1).all() and (m["ncols"] == 3).all()
    same = (stats.T == so).all(axis=1)
    assert same.mean() > 0.9
    assert np.allclose(y[:, same], yo[:, same], rtol=1e-9, atol=1e-13) and np.allclose(y, yo, rtol=1e-5, atol=1e-9)
    assert np.allclose(y[-1][:, 0], cur[:, 0] * 1200.0 / 3600.0, rtol=1e-6)  # discharge capacity = I t
    # full discharge: members with I > ~0.68 A hit V = 3.105 V before t = 3600 s, each at its own time
    t_eval = [600.0, 1800.0, 3600.0]
    y, tot, stats, status, yo, so, failed = run_pair(H, O, "spm", cur, t_eval, 20, group=1, method=0, rtol=1e-6, atol=[1e-6])
    m, ref = run_pair.member, run_pair.oracle_roots
    assert failed == 0 and (status == 0).all()
    assert np.array_equal(m["root_idx"], ref["root_idx"]) and np.array_equal(m["ncols"], ref["ncols"])
    hit = m["root_idx"] >= 0
    assert 10 < hit.sum() < nb and (m["root_idx"][hit] == 0).all()
    assert np.allclose(m["t_root"][hit], ref["t_root"][hit], rtol=1e-6) and np.all(np.diff(m["t_root"][hit]) < 0)  # higher current, earlier cut-off
    ok = np.isfinite(yo)
    assert np.array_equal(np.isfinite(y), ok) and np.allclose(y[ok], yo[ok], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("model,size,n", [("heat1d", 20, 20), ("heat1d", 64, 64), ("robertson_ode", 3, 9), ("gaussian_decay", 12, 12), ("dydt_y2", 10, 10)])
def test_wave_member_other_runtime_sized_models(H, O, model, size, n):
    nb = 12
    rng = np.random.default_rng(n)
    if model == "heat1d":
        p, t_eval, tol = rng.uniform(0.5, 2.0, (nb, 1)), [0.01, 0.1], dict(rtol=1e-6, atol=[1e-6])
    elif model == "robertson_ode":
        p, t_eval, tol = robertson_params(nb), [0.4, 4.0, 40.0], dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6] * 3)
    elif model == "gaussian_decay":
        p, t_eval, tol = rng.uniform(0.5, 2.0, (nb, n)), [0.5, 2.0], dict(rtol=1e-6, atol=[1e-6])
    else:
        p, t_eval, tol = np.zeros((nb, 0)), [0.001, 0.003], dict(rtol=1e-6, atol=[1e-6])
    s = H.Solver(model, p, nbatch=nb, model_size=size, **tol)
    assert s.n == n
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True)
    if p.shape[1] == 0:  # the oracle wants at least a dummy parameter block: one solve serves all (identical) members
        o = O.OracleSolver(ORACLE_MODEL[model], [], model_size=size, **tol)
        o.set_stop_time(t_eval[-1])
        col, ref = 0, np.zeros((len(t_eval), n))
        while col < len(t_eval):
            r = o.step()
            while col < len(t_eval) and t_eval[col] <= o.state()["t"]:
                ref[col] = o.interpolate(t_eval[col])[0]
                col += 1
            if r == 2:
                break
        assert (m["status"] == 0).all() and np.allclose(y, ref[:, None, :], rtol=1e-9, atol=1e-12)
        assert (m["stats"][0] == o.stats()["number_of_steps"]).all()
        return
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL[model], p, t_eval, model_size=size, nthreads=4, **tol)
    yo = np.transpose(yo, (1, 0, 2))
    assert failed == 0 and (m["status"] == 0).all()
    same = (m["stats"].T == so).all(axis=1)
    assert same.mean() > 0.8
    assert np.allclose(y[:, same], yo[:, same], rtol=1e-5, atol=1e-12) and np.allclose(y, yo, rtol=5e-3, atol=1e-9)


# ------------------------------------------------------------------ bit-for-bit verification of the device-side control logic
@pytest.fixture
def det_pow(O):
    O.set_det_pow(True)
    yield
    O.set_det_pow(False)


def _bitwise_pair(H, O, model, p, t_eval, size, group, method, **tol):
    nb = len(p)
    s = H.Solver(model, p, nbatch=nb, model_size=size, method=method, **tol)
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group, deterministic_pow=True)
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL[model], np.asarray(p, dtype=float), t_eval, model_size=size, nthreads=8, group=group, method=method, **tol)
    yo = np.transpose(yo, (1, 0, 2))
    assert failed == 0 and (m["status"] == 0).all()
    assert np.array_equal(m["stats"].T, so), "counters differ"
    assert np.array_equal(y, yo, equal_nan=True), "states differ"
    return m, O.solve_dense_independent.last_roots


@pytest.mark.parametrize("group", [1, 64])
def test_with_a_shared_deterministic_pow_the_resident_bdf_is_bit_identical_to_the_oracle(H, O, det_pow, group):
    """The only arithmetic the device-resident kernels do not share with the oracle is libm's pow().  With include/diffsol_detpow.h on both sides
    (a pow built from IEEE basic operations in a fixed order, within 1 ulp of libm) EVERY member's counters and EVERY output bit must agree: this
    verifies the whole device-side restatement of Bdf::step / Convergence / JacobianUpdate / set_step_size / solve_dense, including the long stress
    horizon t = 4e10 and tight tolerances where libm-vs-ocml runs drift apart."""
    p = robertson_params(700, seed=5)
    _bitwise_pair(H, O, "robertson_ode", p, T_EVAL, 1, group, 0, **ROB)
    _bitwise_pair(H, O, "robertson_ode", p[:200], [0.4 * 10 ** k for k in range(12)], 1, group, 0, **ROB)
    _bitwise_pair(H, O, "robertson_ode", p[:200], T_EVAL[:5], 1, group, 0, rtol=1e-9, atol=[1e-13, 1e-17, 1e-11])
    _bitwise_pair(H, O, "robertson", p[:130], T_EVAL[:5], 0, group, 0, rtol=1e-4, atol=[1e-8, 1e-6, 1e-6])  # DAE: mass matrix + consistent init
    pk = (0.1 * (np.arange(70) + 1))[:, None]
    _bitwise_pair(H, O, "exponential_decay_with_algebraic", pk, [1.0, 5.0], 0, group, 0, rtol=1e-6, atol=[1e-6] * 3)  # inconsistent IC, line search


@pytest.mark.parametrize("group", [1, 64])
@pytest.mark.parametrize("method", [1, 2])
def test_with_a_shared_deterministic_pow_the_resident_sdirk_is_bit_identical_to_the_oracle(H, O, det_pow, method, group):
    p = robertson_params(200, seed=6)
    _bitwise_pair(H, O, "robertson_ode", p, T_EVAL[:5], 1, group, method, **ROB)
    _bitwise_pair(H, O, "robertson", p[:130], [0.4, 4.0, 40.0], 0, group, method, rtol=1e-4, atol=[1e-8, 1e-6, 1e-6])
    k = 0.05 * (np.arange(100) + 1)
    if group == 1:  # per-member events: root times, indices and column counts bitwise too
        m, ref = _bitwise_pair(H, O, "exponential_decay_with_root", np.stack([k, np.ones(100)], axis=1), [0.5, 1.0, 2.0, 4.0, 8.0, 16.0], 0, 1, method,
                               rtol=1e-6, atol=[1e-6, 1e-6])
        assert np.array_equal(m["root_idx"], ref["root_idx"]) and np.array_equal(m["ncols"], ref["ncols"])
        assert np.array_equal(m["t_root"], ref["t_root"], equal_nan=True)


@pytest.mark.parametrize("lane", ["1", "0"])
def test_with_a_shared_deterministic_pow_the_wavefront_per_member_bdf_is_bit_identical_to_the_oracle(H, O, det_pow, monkeypatch, lane):
    """Run-time-sized built-in models, per-member control.  lane = "1" (default): models with a banded Jacobian run the lane-per-member BDF with the state in
    per-lane memory and a banded LU (their static form DynLane<...> of csrc/dsh_models_lane.hpp, instantiated by hiprtc for the size at hand);
    lane = "0" (DSH_RESIDENT_LANE=0): the wavefront-per-member kernel.  Both must give the oracle's bits."""
    monkeypatch.setenv("DSH_RESIDENT_LANE", lane)
    from diffsol_amd import _ffi
    assert _ffi.load_device_lib().dsh_model_lane_twin(H.MODELS["spm"], 20) >= 1000 and _ffi.load_device_lib().dsh_model_lane_twin(H.MODELS["gaussian_decay"], 12) == -1
    cur = np.linspace(0.6, 1.4, 24)[:, None]
    _bitwise_pair(H, O, "spm", cur, [60.0, 600.0, 1200.0], 20, 1, 0, rtol=1e-6, atol=[1e-6])  # before any voltage cut-off
    # full discharge, BASELINE config 4: every member's own cut-off time (the terminal voltage's tanh / asinh / exp are diffsol_detpow.h's on both sides)
    m, ref = _bitwise_pair(H, O, "spm", cur, [600.0, 1800.0, 3600.0], 20, 1, 0, rtol=1e-6, atol=[1e-6])
    assert (m["root_idx"] >= 0).sum() > 10 and np.array_equal(m["root_idx"], ref["root_idx"]) and np.array_equal(m["ncols"], ref["ncols"])
    assert np.array_equal(m["t_root"], ref["t_root"], equal_nan=True)
    rng = np.random.default_rng(1)
    _bitwise_pair(H, O, "heat1d", rng.uniform(0.5, 2.0, (12, 1)), [0.01, 0.1], 64, 1, 0, rtol=1e-6, atol=[1e-6])
    _bitwise_pair(H, O, "robertson_ode", robertson_params(12), [0.4, 4.0, 40.0], 3, 1, 0, rtol=1e-4, atol=[1e-8, 1e-14, 1e-6] * 3)


@pytest.mark.parametrize("method", [1, 2])
def test_with_a_shared_deterministic_pow_the_wavefront_per_member_sdirk_is_bit_identical_to_the_oracle(H, O, det_pow, monkeypatch, method):
    """TR-BDF2 / ESDIRK34 in the wavefront-per-member form (k_sdirk_wave_member: Sdirk::step over Rk with one state component per lane,
    the LU rows in registers, the linearisation requested where the reference resets its Jacobian and carried out before the next Newton solve).  Run-time-
    sized built-in models without a banded twin, and — with DSH_RESIDENT_LANE=0 — the banded ones too, the battery model through its voltage cut-off events:
    states, counters, event times, indices and column counts of every member are the oracle's."""
    _bitwise_pair(H, O, "gaussian_decay", np.random.default_rng(1).uniform(0.5, 2.0, (9, 12)), [0.5, 2.0], 12, 1, method, rtol=1e-6, atol=[1e-6])
    _bitwise_pair(H, O, "robertson_ode", robertson_params(12), [0.4, 4.0, 40.0], 3, 1, method, rtol=1e-4, atol=[1e-8, 1e-14, 1e-6] * 3)
    monkeypatch.setenv("DSH_RESIDENT_LANE", "0")
    rng = np.random.default_rng(1)
    _bitwise_pair(H, O, "heat1d", rng.uniform(0.5, 2.0, (12, 1)), [0.01, 0.1], 64, 1, method, rtol=1e-6, atol=[1e-6])
    cur = np.linspace(0.6, 1.4, 24)[:, None]
    m, ref = _bitwise_pair(H, O, "spm", cur, [600.0, 1800.0, 3600.0], 20, 1, method, rtol=1e-6, atol=[1e-6])
    assert (m["root_idx"] >= 0).sum() > 10 and np.array_equal(m["root_idx"], ref["root_idx"]) and np.array_equal(m["ncols"], ref["ncols"])
    assert np.array_equal(m["t_root"], ref["t_root"], equal_nan=True)


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("group", [1, 64])
def test_banded_models_up_to_512_states_have_a_device_resident_lane_per_member_form(H, O, det_pow, method, group):
    """a device-resident path for 64 < n <= 512 (BASELINE config 3 per member).  The lane-per-member kernels keep the whole solver state
    in per-lane scratch, so the built-in banded models get their static form for any size up to 512 (~230 bytes of scratch per state for BDF; the hardware's
    scratch wave size allows 128 KB per lane).  heat1d at n = 100 (not a multiple of any chunk size), BDF / TR-BDF2 / ESDIRK34, every member its own history
    and wavefront lock-step groups: the oracle's bits.  AUTO keeps small ensembles of such a model on the host-driven path, where they are faster (scripts/heat_resident.py,
    profiles/r02_heat_resident.jsonl: BDF pays from 16 384 members on, TR-BDF2 does not at any size that fits)."""
    from diffsol_amd import _ffi
    assert _ffi.load_device_lib().dsh_model_lane_twin(H.MODELS["heat1d"], 100) >= 1000 and _ffi.load_device_lib().dsh_model_lane_twin(H.MODELS["heat1d"], 513) == -1
    rng = np.random.default_rng(100)
    D = rng.uniform(0.5, 2.0, (70, 1))
    _bitwise_pair(H, O, "heat1d", D, [0.01, 0.1, 0.3], 100, group, method, rtol=1e-6, atol=[1e-6])
    s = H.Solver("heat1d", D, nbatch=70, model_size=100, rtol=1e-6, atol=[1e-6], method=[H.METHOD_BDF, H.METHOD_TR_BDF2, H.METHOD_ESDIRK34][method])
    # AUTO for a small ensemble of such a model: the workgroup-per-member form (64 < n <= 140, dense LU in LDS; round 4: BDF, then TR-BDF2 / ESDIRK34);
    # beyond 140 states every method stays host-driven until the ensemble is large enough for the lane-per-member twin to pay
    assert s.ensemble_mode()[1] == 1
    s2 = H.Solver("heat1d", rng.uniform(0.5, 2.0, (70, 1)), nbatch=70, model_size=200, rtol=1e-6, atol=[1e-6], method=[H.METHOD_BDF, H.METHOD_TR_BDF2, H.METHOD_ESDIRK34][method])
    assert s2.ensemble_mode()[1] == 0


@pytest.mark.parametrize("group", [1, 64])
def test_rlc_config5_esdirk34_with_threshold_events_is_bit_identical_to_the_oracle(H, O, det_pow, group):
    """BASELINE config 5 at reduced size: RLC DAE (singular mass, sin source term), ESDIRK34, root iR - i_thresh.  States, counters, root times,
    root indices and column counts of every member (group=1) / of every 64-member lock-step group (group=64, no events there) bitwise."""
    nb = 200
    rng = np.random.default_rng(5)
    R, Cc = rng.uniform(50.0, 200.0, nb), np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
    thresh = 0.03 if group == 1 else 10.0  # lock-step groups stop as a whole in the host-driven path; the resident group mode runs event-free
    p = np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, thresh)], axis=1)
    m, ref = _bitwise_pair(H, O, "rlc", p, [0.002, 0.005, 0.01, 0.02, 0.05], 1, group, 2, rtol=1e-6, atol=[1e-6] * 4)
    assert np.array_equal(m["root_idx"], ref["root_idx"]) and np.array_equal(m["ncols"], ref["ncols"])
    assert np.array_equal(m["t_root"], ref["t_root"], equal_nan=True)
    if group == 1:
        assert 0 < (m["root_idx"] >= 0).sum() < nb


def _needs_experiments():
    """the measured-slower variants are compiled only with `make EXPERIMENTS=1` (csrc/Makefile, -DDSH_EXPERIMENTS); the shipped library ignores their knobs"""
    from diffsol_amd import _ffi
    if not _ffi.load_device_lib().dsh_experiments_enabled():
        pytest.skip("library built without DSH_EXPERIMENTS (make -C diffsol_amd/csrc EXPERIMENTS=1)")


@pytest.mark.parametrize("var,k", [("DSH_REBIN", "1"), ("DSH_REBIN", "3"), ("DSH_REBIN_STEPS", "16"), ("DSH_REBIN_STEPS", "100")])
def test_segmented_per_member_runs_with_rebinning_between_launches_give_the_bits_of_the_single_launch(H, monkeypatch, var, k):
    """k_bdf_adaptive<.., SEG>: the launch ends for a member after k save points (or k trips of its step loop), the whole per-member integrator state goes to memory,
    the members are dealt to lanes again in another order (device radix sort on (order, steps since the last change, |h|)) and the next launch resumes.  Nothing of the
    arithmetic depends on where a launch ends: every output bit, every counter equals the single launch.  (Measured slower than the single launch in every setting —
    profiles/r03_per_member.md — so it stays an opt-in.)"""
    _needs_experiments()
    p = robertson_params(700, seed=9)
    s = H.Solver("robertson_ode", p, nbatch=len(p), model_size=1, **ROB)
    y0, t0, m0 = s.solve_dense_adaptive(T_EVAL, want_member_stats=True, group=1)
    monkeypatch.setenv(var, k)
    y1, t1, m1 = s.solve_dense_adaptive(T_EVAL, want_member_stats=True, group=1)
    monkeypatch.delenv(var)
    assert np.array_equal(y1, y0) and t1 == t0 and all(np.array_equal(m1[q], m0[q], equal_nan=True) for q in m0)
    assert t0["failed_members"] == 0 and m0["stats"][0].min() < m0["stats"][0].max()


@pytest.mark.parametrize("group", [1, 64])
def test_the_opt_in_fast_arithmetic_variant_stays_within_1e6_relative_of_the_cpu_result(H, O, group):
    """deterministic_pow = 2 (dsh_adaptive_fast.hip: -ffp-contract=fast, reciprocal-math division, ocml pow, reciprocal Newton weights) is the one kernel of the
    library that is not bit-comparable with the oracle.  north_star's bar for it: states within 1e-6 relative of the CPU result — checked at tight tolerances,
    where the integration error itself is below that, against the oracle with libm's pow; at the bench's tolerances the members must still all succeed,
    conserve mass and stay within the solver tolerance of the exact kernel."""
    p = robertson_params(640, seed=3)
    tight = dict(rtol=1e-9, atol=[1e-13, 1e-17, 1e-11])
    s = H.Solver("robertson_ode", p, nbatch=len(p), model_size=1, **tight)
    y, tot = s.solve_dense_adaptive(T_EVAL[:5], group=group, deterministic_pow=2)
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], np.asarray(p, dtype=float), T_EVAL[:5], model_size=1, nthreads=8, group=group, **tight)
    yo = np.transpose(yo, (1, 0, 2))
    assert failed == 0 and tot["failed_members"] == 0
    big = np.abs(yo) > 1e-7
    assert (np.abs(y - yo)[big] / np.abs(yo)[big]).max() < 1e-6
    assert np.allclose(y, yo, rtol=1e-6, atol=1e-13)
    s2 = H.Solver("robertson_ode", p, nbatch=len(p), model_size=1, **ROB)
    yf, totf = s2.solve_dense_adaptive(T_EVAL, group=group, deterministic_pow=2)
    ye, tote = s2.solve_dense_adaptive(T_EVAL, group=group, deterministic_pow=1)
    assert totf["failed_members"] == 0 and np.abs(yf.sum(axis=2) - 1.0).max() < 1e-9
    assert np.allclose(yf, ye, rtol=5e-3, atol=1e-9) and not np.array_equal(yf, ye)
    assert abs(totf["number_of_steps"] - tote["number_of_steps"]) < 0.02 * tote["number_of_steps"]


def test_deterministic_pow_is_the_same_function_on_host_and_device_and_close_to_libm(H, O):
    """diffsol_detpow.h against libm on the host (the device-side identity is what the bitwise tests above establish)."""
    rng = np.random.default_rng(0)
    x, y = np.exp(rng.uniform(-30, 30, 20000)), rng.uniform(-1.3, 1.3, 20000)
    got = np.array([O.det_pow(a, b) for a, b in zip(x, y)])
    ref = np.power(x, y)
    assert np.max(np.abs(got - ref) / np.spacing(ref)) <= 1.0
    assert O.det_pow(0.0, -0.5) == np.inf and O.det_pow(4.0, 0.5) == 2.0 and O.det_pow(7.0, 0.0) == 1.0 and np.isnan(O.det_pow(np.nan, 0.5))


@pytest.mark.parametrize("model,method", [("robertson_ode", 0), ("robertson", 0), ("exponential_decay_with_root", 0), ("exponential_decay_with_algebraic", 0)])
def test_phase_scheduled_per_member_kernel_gives_the_bits_of_the_nested_loop_kernel_and_of_the_oracle(H, O, det_pow, monkeypatch, model, method):
    """Per-member control runs on the nested-loop kernel by default; DSH_MEMBER_SCHED=1 selects the phase-scheduled kernel (k_bdf_member_sched: one
    phase per wavefront pass, chosen by ballot).  Both must give every member's states, counters, event times, column counts and failure codes
    bit for bit — and equal the oracle's independent solves.  Ensembles are sized and parameterised so that lanes of a wavefront sit in different phases
    (different iteration counts, rejected steps, refactorisations, early event stops, a member that fails)."""
    _needs_experiments()
    rng = np.random.default_rng(12)
    nb = 333
    if model in ("robertson_ode", "robertson"):
        p = np.exp(rng.uniform(np.log([0.004, 1e3, 3e6]), np.log([0.4, 1e5, 3e8]), (nb, 3)))
        kw = dict(model_size=1 if model == "robertson_ode" else 0, rtol=1e-5, atol=[1e-9, 1e-13, 1e-8])
        t_eval = [0.4, 4.0, 40.0, 400.0, 4e3, 4e4]
    elif model == "exponential_decay_with_root":
        p = np.stack([rng.uniform(0.01, 5.0, nb), rng.uniform(0.7, 3.0, nb)], axis=1)
        kw = dict(model_size=0, rtol=1e-7, atol=[1e-8, 1e-8])
        t_eval = [0.1, 0.5, 2.0, 8.0, 30.0]
    else:
        p = rng.uniform(0.05, 5.0, (nb, 1))
        kw = dict(model_size=0, rtol=1e-6, atol=[1e-7] * 3)
        t_eval = [0.5, 3.0, 20.0]
    out = {}
    for sched in ("1", "0"):
        monkeypatch.setenv("DSH_MEMBER_SCHED", sched)
        s = H.Solver(model, p, nbatch=nb, method=method, **kw)
        out[sched] = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    (y1, tot1, m1), (y0, tot0, m0) = out["1"], out["0"]
    assert np.array_equal(y1, y0, equal_nan=True) and tot1 == tot0
    for k in ("stats", "status", "root_idx", "ncols"):
        assert np.array_equal(m1[k], m0[k]), k
    assert np.array_equal(m1["t_root"], m0["t_root"], equal_nan=True)
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL[model], p, t_eval, nthreads=8, method=method, **kw)
    ok = m1["status"] == 0
    assert int((~ok).sum()) == failed
    assert np.array_equal(y1[:, ok], np.transpose(yo, (1, 0, 2))[:, ok], equal_nan=True) and np.array_equal(m1["stats"].T[ok], so[ok])
    # a run in which members fail (error-test limit): the same failure codes from both kernels
    for sched in ("1", "0"):
        monkeypatch.setenv("DSH_MEMBER_SCHED", sched)
        s = H.Solver(model, p, nbatch=nb, method=method, options=dict(max_error_test_failures=1, max_nonlinear_solver_failures=2), **kw)
        out[sched] = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    assert np.array_equal(out["1"][2]["status"], out["0"][2]["status"]) and np.array_equal(out["1"][0], out["0"][0], equal_nan=True)
    assert np.array_equal(out["1"][2]["stats"], out["0"][2]["stats"])


@pytest.mark.parametrize("group", [1, 64])
def test_streaming_lane_per_member_kernel_gives_the_bits_of_the_register_array_form(H, monkeypatch, group):
    """Banded run-time-sized models run k_bdf_lane_banded (dsh_lane_banded_kernel.hpp: fused passes over per-lane memory, chunked loads, buffer-index swap
    of diff / diff_tmp, next prediction made by the accept pass); DSH_LANE_BANDED_V1=1 selects k_bdf_adaptive's banded branch (one loop per vector
    operation).  Same arithmetic in the same order: states, counters, event data bit for bit — for sizes that are / are not multiples of the load chunks
    (13 is prime), bandwidth 1 and 2, per-member control and wavefront lock-step groups, with steps that fail and orders that change."""
    _needs_experiments()
    rng = np.random.default_rng(8)
    cases = [("heat1d", rng.uniform(0.5, 2.0, (130, 1)), [0.01, 0.1, 0.3], 13, dict(rtol=1e-6, atol=[1e-7])),
             ("robertson_ode", robertson_params(70), [0.4, 4.0, 40.0, 400.0], 4, dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6] * 4)),
             ("spm", np.linspace(0.6, 1.4, 70)[:, None], [600.0, 1800.0, 3600.0], 20, dict(rtol=1e-6, atol=[1e-6]))]
    for model, p, t_eval, size, tol in cases:
        if group == 64 and model == "spm":
            t_eval = [60.0, 600.0, 1200.0]  # before any cut-off: lock-step groups have no per-member events
        out = {}
        for v1 in ("0", "1"):
            monkeypatch.setenv("DSH_LANE_BANDED_V1", v1)
            s = H.Solver(model, p, nbatch=len(p), model_size=size, **tol)
            out[v1] = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group)
        (ya, ta, ma), (yb, tb, mb) = out["0"], out["1"]
        assert ta == tb and ta["number_of_steps"] > 20 * len(p), model
        assert np.array_equal(ya, yb, equal_nan=True), model
        for k in ma:
            assert np.array_equal(ma[k], mb[k], equal_nan=True), (model, k)


@pytest.mark.parametrize("model", ["robertson_ode", "spm", "rlc"])
def test_per_member_ensembles_run_sorted_by_parameters_and_return_every_result_in_the_callers_order(H, monkeypatch, model):
    """Per-member device-resident solves of >= 1024 members run the ensemble along a Z-order curve through parameter space (host/solver_c.cpp
    prepare_member_order: neighbours in a wavefront have similar parameters, hence similar step counts and paths) and un-permute every output.  A member's
    result does not depend on its position: states, per-member counters, status, event times / indices / column counts and the totals are bit for bit those
    of the unsorted run (DSH_MEMBER_SORT=0) — BDF on registers, BDF on per-lane memory with events, ESDIRK34 on a DAE with events."""
    rng = np.random.default_rng(21)
    nb = 2500
    if model == "robertson_ode":
        p, kw, t_eval = robertson_params(nb, seed=4), dict(model_size=1, rtol=1e-4, atol=[1e-8, 1e-14, 1e-6]), [0.4, 4.0, 40.0, 400.0]
    elif model == "spm":
        p, kw, t_eval = rng.uniform(0.6, 1.4, (nb, 1)), dict(model_size=20, rtol=1e-6, atol=[1e-6]), [600.0, 1800.0, 3600.0]
    else:
        R, Cc = rng.uniform(50.0, 200.0, nb), np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
        p = np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, 0.03)], axis=1)
        kw, t_eval = dict(method=2, rtol=1e-6, atol=[1e-6] * 4), [0.002, 0.005, 0.01, 0.02, 0.05]
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DSH_MEMBER_SORT", flag)
        s = H.Solver(model, p, nbatch=nb, **kw)
        out[flag] = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    (ya, ta, ma), (yb, tb, mb) = out["1"], out["0"]
    assert ta == tb and np.array_equal(ya, yb, equal_nan=True)
    for k in ma:
        assert np.array_equal(ma[k], mb[k], equal_nan=True), k
    assert len(np.unique(ma["stats"][0])) > 3  # members really differ
