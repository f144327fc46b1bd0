"""oracle.solve_dense_independent_sens (solve_dense_sensitivities per member, sensitivities.rs:114-260 over bdf.rs:934-989): the checker of the device-resident
BDF with forward sensitivities.  Pinned here against what it must equal: the oracle's own stepping solver with sensitivities (whose 13 counters reproduce the
reference's sensitivity snapshots, tests/test_oracle_golden.py), analytic derivatives, and central differences."""
import numpy as np

from helpers import ORACLE_MODEL


def test_member_solves_with_sensitivities_equal_the_stepping_solver_and_the_analytic_derivatives(O):
    k, y0 = np.array([0.1, 0.3, 0.7]), np.array([1.0, 2.0, 0.5])
    p = np.stack([k, y0], axis=1)
    te = [0.0, 1.0, 2.5, 10.0]
    tol = dict(rtol=1e-6, atol=[1e-6, 1e-6])
    y, s, st, failed = O.solve_dense_independent_sens(ORACLE_MODEL["exponential_decay"], p, te, sens_rtol=1e-6, sens_atol=[1e-6], **tol)
    assert failed == 0 and y.shape == (3, 4, 2) and s.shape == (2, 3, 4, 2)
    t = np.asarray(te)[None, :]
    e = np.exp(-k[:, None] * t)
    assert np.allclose(y[:, :, 0], y0[:, None] * e, rtol=2e-4, atol=1e-6)
    assert np.allclose(s[0, :, :, 0], -t * y0[:, None] * e, rtol=1e-3, atol=3e-5) and np.allclose(s[1, :, :, 0], e, rtol=1e-3, atol=3e-5)
    # member 0 is the reference's exponential_decay_problem_sens: the same steps as the stepping solver integrating to the last save point
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay"], p[0], sens=True, sens_rtol=1e-6, sens_atol=[1e-6], **tol)
    o.solve(te[-1])
    stt = o.stats()
    assert (st[0, 0], st[0, 1], st[0, 2]) == (stt["number_of_steps"], stt["number_of_nonlinear_solver_iterations"], stt["number_of_linear_solver_setups"])
    assert np.array_equal(o.interpolate_sens(te[-1])[:, 0, :], s[:, 0, -1, :])
    # a lock-step group of all three members is ONE batched problem: different steps from the member solves, same derivatives within tolerance
    yg, sg, stg, fg = O.solve_dense_independent_sens(ORACLE_MODEL["exponential_decay"], p, te, group=3, sens_rtol=1e-6, sens_atol=[1e-6], **tol)
    assert fg == 0 and (stg[0] == stg[1]).all() and np.allclose(sg, s, rtol=1e-3, atol=3e-5)


def test_robertson_sensitivities_are_the_central_differences_of_the_states(O):
    p = np.array([[0.04, 1e4, 3e7], [0.05, 2e4, 2e7]])
    te = [0.4, 4.0, 40.0]
    tol = dict(rtol=1e-7, atol=[1e-10, 1e-12, 1e-10])
    y, s, st, failed = O.solve_dense_independent_sens(ORACLE_MODEL["robertson_ode"], p, te, model_size=1, **tol)
    y0, st0, f0 = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], p, te, model_size=1, **tol)
    assert failed == 0 and f0 == 0 and np.allclose(y, y0, rtol=1e-5, atol=1e-12)
    h = 1e-4
    for j in range(3):
        pp, pm = p.copy(), p.copy()
        pp[:, j] *= 1 + h
        pm[:, j] *= 1 - h
        yp, _, _ = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], pp, te, model_size=1, **tol)
        ym, _, _ = O.solve_dense_independent(ORACLE_MODEL["robertson_ode"], pm, te, model_size=1, **tol)
        fd = (yp - ym) / (2 * h * p[:, j])[:, None, None]
        assert np.abs(s[j] - fd).max() < (2e-3 if j < 2 else 0.1) * np.abs(fd).max()  # dy/dp_3 ~ 3e-9: the differences carry the states' own 1e-10 errors
